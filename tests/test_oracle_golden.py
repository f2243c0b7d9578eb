"""CPU: the in-repo oracle against the golden vectors dumped from the unmodified reference
(tests/golden/*.npz, generator tests/golden/make_golden.py). Exact equality, every bit."""
import numpy as np
import pytest

import golden_util as gu


@pytest.mark.parametrize("name", sorted(gu.CASES))
def test_oracle_matches_reference_dump(po, name):
    c = gu.CASES[name]
    r, x1, x2 = gu.inputs_m(name)
    ora = po.Oracle(c["lx"], c["ly"], r, x1, x2)
    g = gu.load(name)
    s = ora.scalars()
    assert np.array_equal(np.array([s[k] for k in po.SCALARS], float)[:6], g["scalars"][:6]), "time-step derivation"
    res = gu.run_case(ora, name)
    gu.compare(name, res)
    s = ora.scalars()
    assert np.array_equal(np.array([s[k] for k in po.SCALARS], float), g["scalars"])


@pytest.mark.parametrize("name,dumps", [("real_a08d83_600x500", None), ("real_7000_2048x2048", (1, 2)),
                                        ("real_50000_4096x4096", (1,)), ("real_50000test_3072x3072", (1,)),
                                        ("real_50000_8192x4096", (1,)),
                                        # the JUBE cases of the reference's benchmark.xml and the remaining samples
                                        ("real_a08d83_2000x1000", None), ("real_a6d83_2000x1000", None),
                                        ("real_spl04_1600x2400", (1, 2)), ("real_slope_12500x2500", (1,))])
def test_oracle_on_the_reference_s_own_samples(po, name, dumps):
    """The reference's shipped inputs at BASELINE.json's sizes (a08d83 @ 600x500; a08_a4b4r18_7000 @ 2048^2 =
    configs[2]; 50000.data @ 4096^2 = configs[3]; 50000-test.data @ 3072^2 = configs[0], the CPU-only plumbing case;
    50000.data @ 8192x4096 = configs[4] as one domain): the state after whole coupled steps must hash to what the
    unmodified reference produced (tests/golden/real_*.npz). The serial total density too (same order).
    Round 3: the reference's own benchmark cases (benchmark.xml:7-27) -- a08d83 and a6d83 @ 2000x1000 (a6d83's packing
    is 1493 nodes tall: a third of its grains lie OUTSIDE the lattice, clipped by the bounding-box clamps main.c:1016-1023,
    1300-1303) and slope @ 12500x2500 (radii up to 1.5 mm) -- and spl04.data. The CPU suite checks the first dumps (the
    oracle needs ~6 s per fluid step at 12500x2500); the GPU suite checks every dump of every case (10-20 fluid steps)."""
    class Sim:
        def __init__(self, lx, ly, r, x1, x2): self.o = po.Oracle(lx, ly, r, x1, x2)
        def steps(self, n): self.o.steps(n)
    gu.check_real_case(name, Sim, lambda s: (s.o.get_f(), s.o.get_obst(), s.o.get_fhf(), s.o.get_grains()[:, :9],
                                             s.o.total_density()), dumps)


def test_known_answer_scalars_of_the_real_samples(po):
    """Derived run constants the survey measured on the reference for its own sample geometry
    (SURVEY.md section 8c / BASELINE.md section 2): they depend only on lx, scale and the smallest radius."""
    # a08d83.data: r_min = 0.61 mm -> npDEM 10 at 600x500; 50000.data: r_min 0.5 mm -> npDEM 12 at 4096^2
    o = po.Oracle(600, 500, [0.61e-3], [3e-3], [3e-3]).scalars()
    assert o["npDEM"] == 10 and abs(o["dx"] - 1.001669e-04) < 1e-10 and abs(o["c"] - 7.4875) < 1e-9
    o = po.Oracle(4096, 64, [0.5e-3], [3e-3], [3e-3]).scalars()
    assert o["npDEM"] == 12 and abs(o["dx"] - 1.000244e-04) < 1e-10 and abs(o["c"] - 7.498169) < 1e-6
    assert abs(o["dtLB"] - 1.333985e-05) < 1e-11 and abs(o["dt"] - 1.111654e-06) < 1e-12


def test_sample_reader_formats(po, tmp_path):
    """read_sample (main.c:609-639): comment line, count, rows with tabs/spaces, optional ';'."""
    p = tmp_path / "s.data"
    p.write_text("#comment line\n3\n1.0\t2.0\t3.0\n0.5 4.0 5.0;\n7.5e-1\t6\t7;\n")
    r, x1, x2 = po.read_sample(str(p))
    assert np.array_equal(r, np.array([1.0, 0.5, 0.75]) * 1e-3)
    assert np.array_equal(x1, np.array([2.0, 4.0, 6.0]) * 1e-3)
    assert np.array_equal(x2, np.array([3.0, 5.0, 7.0]) * 1e-3)
    with pytest.raises(RuntimeError):
        po.read_sample(str(tmp_path / "missing.data"))
    q = tmp_path / "short.data"
    q.write_text("#c\n3\n1 2 3\n")
    with pytest.raises(RuntimeError):
        po.read_sample(str(q))


def test_swap_stream_is_a_pull(po):
    """SURVEY.md Notes: the two-pass swap streaming equals f_new[P][q] = f*[P-e_q][q], with the
    opposite population kept at the array edge. Checked on the oracle with numpy."""
    lx, ly = 40, 33
    ora = po.Oracle(lx, ly, [0.6e-3], [2e-3], [1.6e-3])
    rng = np.random.default_rng(0)
    f0 = rng.random((lx, ly, 9))
    ora.set_f(f0)
    ora.swap_stream()
    ex = [0, -1, -1, -1, 0, 1, 1, 1, 0]; ey = [0, 1, 0, -1, -1, -1, 0, 1, 1]
    want = np.empty_like(f0)
    want[..., 0] = f0[..., 0]
    for q in range(1, 9):
        qo = q + 4 if q <= 4 else q - 4
        for x in range(lx):
            for y in range(ly):
                sx, sy = x - ex[q], y - ey[q]
                want[x, y, q] = f0[sx, sy, q] if (0 <= sx < lx and 0 <= sy < ly) else f0[x, y, qo]
    assert np.array_equal(ora.get_f(), want)

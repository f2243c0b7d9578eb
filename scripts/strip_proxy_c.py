"""8-GPU projection of the C transport's period from ONE GPU (no multi-GPU box is available to the builder).

The bench workload (4096^2, 50 000 grains; or 8192 x 4096) is cut into `world` strips, ONE PROCESS PER STRIP, all on
this GPU, talking through the tests' RCCL stand-in (tests/rccl_shim: real RCCL refuses several ranks per device). After a
lock-step warm-up with real messages (lbmdem_comm_run on every rank), the other ranks go idle and ONE interior rank is
stepped alone and timed through the very same lbmdem_comm_run -- host enqueue, stream hand-overs, its fluid kernels on its
rows, its link tables, its owned + margin grains -- with the transfers themselves switched off (experiment build:
lbmdem_comm_debug_stale; the buffers and halo rows keep what arrived last: same sizes, same work). The two exchanges on
the critical path are priced separately with the real RCCL (scripts/exchange_probe.py, as in scripts/strip_proxy.py).

usage: python scripts/strip_proxy_c.py [world] [lx]          (needs make AB=1; run by scripts/strip_proxy_c.sh)
Prints one JSON line."""
import json, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "scripts")]
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
AB = os.path.join(ROOT, "2d-lbm-dem_amd", "liblbmdem_hip_ab.so")
SHIM = os.path.join(ROOT, "tests", "rccl_shim", "librccl.so.1")
PERIODS = 100


def worker(rank, world, lx, idfile, outfile):
    import numpy as np
    import __graft_entry__ as ge, samples
    pkg = ge.load_package(); strips = pkg.strips_module()
    ly = 4096
    r, x, y = samples.row_packing(lx, ly, 50000, seed=1234); r, x1, x2 = samples.to_metres(r, x, y)
    if rank == 0:
        uid = pkg.comm_unique_id()
        open(idfile + ".tmp", "wb").write(uid); os.rename(idfile + ".tmp", idfile)
    else:
        while not os.path.exists(idfile):
            time.sleep(0.01)
        uid = open(idfile, "rb").read()
    sim = pkg.LbmDem(lx, ly, r, x1, x2, device=0, strip=strips.partition(lx, world)[rank], halo=2)
    sim.dist_enable(0)
    comm = pkg.Comm(uid, rank, world, 0)
    npdem = sim.cfg.npDEM
    comm.run(sim, 3 * npdem); sim.sync()          # lock-step warm-up: real messages between all ranks
    comm.allreduce_sum(np.zeros(1))               # everybody is through
    if rank != world // 2:        # leave the GPU to the timed rank: several processes' queues are time-sliced by the scheduler
        comm.close(); sim.close()
        open(outfile + f".gone{rank}", "w").close()
        os._exit(0)
    t0 = time.time()
    while sum(os.path.exists(outfile + f".gone{k}") for k in range(world)) < world - 1 and time.time() - t0 < 60:
        time.sleep(0.05)
    time.sleep(1.0)               # ... and their contexts are torn down
    L = pkg.load_library()
    assert L.lbmdem_comm_debug_stale(comm._c, 1) == 0
    def sync():
        try:
            sim.sync()
        except pkg.LbmDemError:   # stale neighbour messages no longer fit the moved grains exactly (flagged, harmless here)
            pass
    comm.run(sim, 5 * npdem); sync()
    t0 = time.perf_counter(); comm.run(sim, PERIODS * npdem); t_enq = time.perf_counter() - t0
    sync()
    ms = 1e3 * (time.perf_counter() - t0) / PERIODS
    # The host floor (task 2a): 100 periods enqueued back to back return only as the GPU works them off (the
    # queue fills up), so their enqueue time equals the GPU time whatever the host costs. Two periods at a time, onto an idle
    # stream, the call returns when the HOST is done: its launches, event operations and RCCL groups per period.
    host_only = []
    for _ in range(30):
        sync()
        t0 = time.perf_counter(); comm.run(sim, 2 * npdem); host_only.append(1e3 * (time.perf_counter() - t0) / 2)
    sync()
    host_only.sort()
    a, g = sim.force_stats()
    json.dump({"rank": rank, "pid": os.getpid(), "alone_ms_per_step": round(ms, 4), "host_enqueue_ms_per_step": round(1e3 * t_enq / PERIODS, 4),
               "host_only_ms_per_period_median_of_30_pairs_on_an_idle_stream": round(host_only[len(host_only) // 2], 4),
               "host_only_ms_per_period_min": round(host_only[0], 4),
               "forces_from_table_and_gathered": [int(a), int(g)],
               "messages_doubles": {k: sim.dist_message_doubles(v) for k, v in (("kin", 0), ("fhf", 1), ("tables", 2))}},
              open(outfile, "w"))


def main():
    world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    lx = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
    import tempfile
    tmp = tempfile.mkdtemp()
    env = dict(os.environ, LBMDEM_HIP_LIBRARY=AB, LBMDEM_RCCL_LIBRARY=SHIM, RCCL_SHIM_TIMEOUT_S="120", LBMDEM_IGNORE_DIST_ERRORS="1")
    procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--worker", str(k), str(world), str(lx),
                               os.path.join(tmp, "id"), os.path.join(tmp, "out.json")], env=env) for k in range(world)]
    rcs = [p.wait() for p in procs]
    assert rcs == [0] * world, rcs
    mid = json.load(open(os.path.join(tmp, "out.json")))
    # the single-GPU step and the price of an exchange, with the real RCCL (one-rank communicator), in this process
    os.environ["LBMDEM_HIP_LIBRARY"] = AB
    import torch  # noqa: F401
    import __graft_entry__ as ge, samples
    from exchange_probe import exchange_probe
    pkg = ge.load_package(); strips = pkg.strips_module()
    ly = 4096
    r, x, y = samples.row_packing(lx, ly, 50000, seed=1234); r, x1, x2 = samples.to_metres(r, x, y)
    cfg = pkg.derive(lx, ly, r)
    out = {"workload": f"{lx}x{ly} / {len(r)} grains, {world} strips of {lx // world} rows (one process each, C transport), halo 2",
           "driver": "lbmdem_comm_run", "variant": "edge rows on the main stream" if os.environ.get("LBMDEM_COMM_EDGES_MAIN") else "product"}
    single = pkg.LbmDem(lx, ly, r, x1, x2)
    single.renderScene(20 * cfg.npDEM); single.sync()
    t0 = time.perf_counter(); single.renderScene(100 * cfg.npDEM); single.sync()
    out["one_gpu_ms_per_step"] = round(1e3 * (time.perf_counter() - t0) / 100, 4)
    del single
    out["mid_rank"] = mid
    comm = pkg.Comm(pkg.comm_unique_id(), 0, 1, 0)
    lat = {}
    for name in ("tables", "fhf"):
        w, wo, inline = exchange_probe(pkg, comm, mid["messages_doubles"][name], 300)
        lat[name] = round(min(w, inline) - wo, 1)
        lat[name + "_loop_side_stream_without_inline_us"] = [round(w, 1), round(wo, 1), round(inline, 1)]
    out["rccl_self_exchange_us"] = lat
    out["projected_ms_per_step"] = round(mid["alone_ms_per_step"] + 1e-3 * (lat["tables"] + lat["fhf"]), 4)
    out["projected_speedup"] = round(out["one_gpu_ms_per_step"] / out["projected_ms_per_step"], 2)
    out["note"] = ("projection = an interior rank alone through lbmdem_comm_run (transfers off) + the two exchanges on its critical "
                   "path (tables, forces) at the measured self-exchange time of the real RCCL (one-rank communicator: hand-overs + "
                   "RCCL kernel, no wire); the kinematics and the f halo overlap the fluid kernels")
    print(json.dumps(out))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--worker":
        worker(int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), sys.argv[5], sys.argv[6])
    else:
        main()

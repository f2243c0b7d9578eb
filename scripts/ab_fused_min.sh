# interleaved A/B of library builds on ONE GPU, fused-kernel time: REPS rounds over all builds, then min / median per build
# usage: bash scripts/ab_fused_min.sh <suffix> ...   (liblbmdem_hip<suffix>.so; REPS, STEPS from the env)
export LBMDEM_BENCH_NO_LEGS=1   # bench.py: no 200-step / real50k legs behind the timed region
cd $GRAFT_REPO_ROOT
P='import sys,json; d=json.loads(sys.stdin.read()); print(d["collide_stream_kernel_ms"], d["ms_per_step"])'
rm -f /tmp/ab_*.txt
for rep in $(seq ${REPS:-6}); do for lib in "$@"; do
  LBMDEM_HIP_LIBRARY=$PWD/2d-lbm-dem_amd/liblbmdem_hip$lib.so python bench.py --steps ${STEPS:-40} --warmup 5 --settle ${SETTLE:-20} --no-cpu-baseline 2>/dev/null | tail -1 | python -c "$P" >> /tmp/ab_$lib.txt
done; done
for lib in "$@"; do python - "$lib" <<'PY'
import sys
lib=sys.argv[1]
rows=[tuple(map(float,l.split())) for l in open(f"/tmp/ab_{lib}.txt") if l.strip()]
k=sorted(r[0] for r in rows); s=sorted(r[1] for r in rows)
print(f"[{lib}] fused min {k[0]:.4f} med {k[len(k)//2]:.4f} max {k[-1]:.4f} | step min {s[0]:.4f} med {s[len(s)//2]:.4f}  (n={len(k)})")
PY
done

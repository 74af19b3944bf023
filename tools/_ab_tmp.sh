R=$GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_dense.py -q -x -k "prologue" 2>&1 | tail -2
for round in 1 2; do
  echo -n "nofuse: "; SBEV_NO_LN_FUSE=1 python bench.py --no-cpu-baseline --no-alt 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'])"
  echo -n "fused: "; python bench.py --no-cpu-baseline --no-alt 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'])"
done
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/ab_fused -o bench -- python $R/bench.py --no-cpu-baseline --no-alt --steps 20 > /dev/null 2>&1
grep "small_kernel\|reduce_kernel\|pair_kernel" $R/gpurun_out/ab_fused/bench_kernel_stats.csv | cut -d, -f1-4 | cut -c30-150

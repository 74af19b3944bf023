python -m pytest tests/test_gpu_sampling.py tests/test_gpu_decoder.py -x -q 2>&1 | tail -2
for rep in 1 2; do python tools/bench_sampler.py --layout mix | tail -1 | cut -c1-120; done
python tools/bench_sampler.py --layout ref | tail -1 | cut -c1-120

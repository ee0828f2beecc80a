mkdir -p gpurun_out/ab
python -m pytest tests -m gpu -x -q 2>&1 | tail -5 > gpurun_out/ab/test.log
python bench.py --steps 30 --warmup 5 --profile-all --no-cpu-baseline --no-full-run > gpurun_out/ab/bench_on.json 2> gpurun_out/ab/bench_on.err
python bench.py --steps 30 --warmup 5 --profile-all --no-cpu-baseline --no-full-run --option edge_key_split=0 > gpurun_out/ab/bench_off.json 2> gpurun_out/ab/bench_off.err
python bench.py --workload c1 --no-cpu-baseline --profile-all > gpurun_out/ab/bench_c1.json 2> gpurun_out/ab/bench_c1.err

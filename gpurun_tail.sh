python -c "import __graft_entry__ as g; g.build(); g.smoke()" 2>&1 | tail -2
python -m pytest tests -m gpu -q 2>&1 | tail -6 | cut -c1-250
python bench.py > gpurun_out/r6_ac_bench_default.json 2> gpurun_out/r6_ac_bench_default.err; head -c 400 gpurun_out/r6_ac_bench_default.json

cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r01b_stats -- python3 bench.py --steps 5 --warmup 1 > gpurun_out/r01b_bench_under_rocprof.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/r01b_pmc_fetch -- python3 tools/pmc_kernels.py > gpurun_out/r01b_pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/r01b_pmc_write -- python3 tools/pmc_kernels.py > gpurun_out/r01b_pmc_write.log 2>&1
python bench.py > gpurun_out/r01b_bench_default.json 2> gpurun_out/r01b_bench_default.err
ls gpurun_out/r01b_*/*/ | head -30
tail -c 600 gpurun_out/r01b_bench_default.json

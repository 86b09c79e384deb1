# Round profile (run on the GPU box through gpurun): kernel stats of the whole bench, PMC traffic of the two roofline
# kernels (separate --pmc passes, --kernel-trace only), the per-layer timeline of the temporal stack, the kernel make-up of
# the two deployed ticks (B = 1: classifier-free and classifier guidance), the default bench.
R=${R:-r06}
export HIP_FORCE_DEV_KERNARG=1     # as the package sets it at import (under the profiler the GPU is initialised before Python starts)
rm -rf gpurun_out/${R}_stats gpurun_out/${R}_stats2 gpurun_out/${R}_pmc_fetch gpurun_out/${R}_pmc_write gpurun_out/${R}_tconv_trace
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
# kernel stats of the bench: ONE stream for the perception pass (ADX_RESNET_STREAMS=1 ADX_PERCEPTION_AHEAD=0), so that a kernel's
# duration in the summary is its own (bench.py's roofline.avg_launch_ms_rocprof); then the same command as the product runs it
# (two sub-batch streams + the pass stream: concurrent kernels share the chip and their durations overlap)
ADX_RESNET_STREAMS=1 ADX_PERCEPTION_AHEAD=0 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${R}_stats -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-deployed > gpurun_out/${R}_bench_under_rocprof.json 2> gpurun_out/${R}_bench_under_rocprof.err
rm -rf gpurun_out/${R}_stats2
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${R}_stats2 -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-deployed --no-train > gpurun_out/${R}_bench_under_rocprof_two_streams.json 2> gpurun_out/${R}_bench_under_rocprof_two_streams.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/${R}_pmc_fetch -- python3 tools/pmc_kernels.py > gpurun_out/${R}_pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/${R}_pmc_write -- python3 tools/pmc_kernels.py > gpurun_out/${R}_pmc_write.log 2>&1
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/${R}_tconv_trace -- python3 tools/bench_tconv.py > gpurun_out/${R}_tconv_layers_hostclock.log 2>&1
python tools/trace_chunks.py gpurun_out/${R}_tconv_trace > gpurun_out/${R}_tconv_layers.txt
rm -rf gpurun_out/${R}_tick_free gpurun_out/${R}_tick_cls
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/${R}_tick_free -- python3 tools/tick_timeline.py run > gpurun_out/${R}_tick_run.log 2>&1
MODE=classifier rocprofv3 --kernel-trace --output-format csv -d gpurun_out/${R}_tick_cls -- python3 tools/tick_timeline.py run >> gpurun_out/${R}_tick_run.log 2>&1
python tools/tick_timeline.py analyze gpurun_out/${R}_tick_free > gpurun_out/${R}_tick_free_b1.txt
python tools/tick_timeline.py analyze gpurun_out/${R}_tick_cls > gpurun_out/${R}_tick_classifier_b1.txt
# the training step: kernel table, the activation-layout levels A/B, idle time of the traced step, host clock
rm -rf gpurun_out/prof_tr gpurun_out/${R}_tr_trace
bash tools/profile_train.sh > gpurun_out/${R}_train_step_kernels.txt 2>&1
cp $(ls -t gpurun_out/prof_tr/*/*kernel_stats.csv | head -1) gpurun_out/${R}_train_kernel_stats.csv
[ -n "$AB_TRAIN_CELLS" ] && python tools/ab_env_train.py ADX_TRAIN_CELLS 0 1 2 3 4 5 > gpurun_out/${R}_ab_train_cells.txt 2>&1
NOSYNC=1 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/${R}_tr_trace -- python3 tools/train_time.py > /dev/null 2>&1
python tools/train_gaps.py gpurun_out/${R}_tr_trace > gpurun_out/${R}_train_gaps.txt 2>&1
rm -rf gpurun_out/${R}_tr_trace
python3 tools/host_times.py 2>/dev/null | tail -22 > gpurun_out/${R}_train_host_times.txt
python bench.py > gpurun_out/${R}_bench_default.json 2> gpurun_out/${R}_bench_default.err
ls gpurun_out/${R}_*/*/ | head -30
tail -c 400 gpurun_out/${R}_bench_default.json

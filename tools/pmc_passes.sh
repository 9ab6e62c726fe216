set -u
# counters are collected with EU_WF_STREAMS=1: with two concurrent band pipelines the per-dispatch counters of overlapping kernels mix
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/pmc
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $R/gpurun_out/pmc/counters_list.txt 2>&1
run() { name=$1; shift; EU_WF_STREAMS=1 timeout -k 10 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $R/gpurun_out/pmc/$name -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/pmc/$name.log 2>&1; echo "$name rc=$?"; }
run a SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU
run b SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_IFETCH SQ_INSTS_FLAT SQ_THREAD_CYCLES_VALU
run c FETCH_SIZE
run d WRITE_SIZE
run e GRBM_GUI_ACTIVE

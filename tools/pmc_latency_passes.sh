set -u
# Where do the waves wait?  Instruction-cache / scalar-cache hit rates and the average latency of scalar, LDS and vector memory
# instructions (LEVEL counters accumulate "instructions in flight" per cycle: LEVEL / INSTS = average latency in cycles).
# Single band stream, as in pmc_passes.sh.  Usage: [BENCH_ARGS=...] [PMC_DIR=pmc_lat] tools/pmc_latency_passes.sh
R=$GRAFT_REPO_ROOT
PMC_DIR=${PMC_DIR:-pmc_lat}
rm -rf $R/gpurun_out/$PMC_DIR
mkdir -p $R/gpurun_out/$PMC_DIR
cd /tmp && export TMPDIR=/tmp
run() { name=$1; shift; EU_WF_STREAMS=${STREAMS:-1} timeout -k 10 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $R/gpurun_out/$PMC_DIR/$name -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-other-configs ${BENCH_ARGS:-} > $R/gpurun_out/$PMC_DIR/$name.log 2>&1; echo "$name rc=$?"; }
run f SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQC_DCACHE_MISSES_DUPLICATE
run g SQ_INST_LEVEL_SMEM SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES
run h SQ_INST_LEVEL_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
run i SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_FLAT
run j SQ_IFETCH_LEVEL SQ_IFETCH SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY
run k SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_INST_CYCLES_SMEM SQ_ACTIVE_INST_VALU
# per-kernel sums
python3 - $R/gpurun_out/$PMC_DIR <<'PY'
import csv, glob, collections, sys, re
root = sys.argv[1]
tot = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob(root + "/*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "")
        tot[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k][r["Counter_Name"]] += 1
with open(root + "/summary.txt", "w") as o:
    for k in sorted(tot):
        if "eu_" not in k: continue
        o.write(k + "\n")
        for c in sorted(tot[k]): o.write("  %-32s launches %5d  avg %16.1f\n" % (c, n[k][c], tot[k][c] / n[k][c]))
print(open(root + "/summary.txt").read())
PY

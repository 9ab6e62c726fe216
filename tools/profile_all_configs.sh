set -u
cd $GRAFT_REPO_ROOT
for cfg in "room|--scene 3d_room.json --max-depth 8" "hallways|--scene 3d_hallways.json --max-depth 12" "4dframe|--scene 4d_frame.json --max-depth 8"; do
  tag=${cfg%%|*}; args=${cfg#*|}
  BENCH_ARGS="$args" bash tools/kernel_stats.sh r02_$tag > gpurun_out/r02_${tag}_kernel_stats.txt 2>&1
  BENCH_ARGS="$args" bash tools/kernel_stats.sh r02_${tag}_1s EU_WF_STREAMS=1 > gpurun_out/r02_${tag}_kernel_stats_single_stream.txt 2>&1
  BENCH_ARGS="$args" bash tools/frame_dispatches.sh r02_$tag EU_WF_STREAMS=1 > gpurun_out/r02_${tag}_dispatches.txt 2>&1
  BENCH_ARGS="$args" PMC_DIR=pmc_$tag bash tools/pmc_passes.sh > gpurun_out/r02_${tag}_pmc_passes.log 2>&1
  echo "done $tag"
done

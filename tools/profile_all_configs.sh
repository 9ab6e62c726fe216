set -u
# kernel statistics (default streams and one band stream), the dispatch trace of one frame and the PMC passes of the three BASELINE
# single-GPU configurations, specialised kernels.  Usage (GPU box): bash tools/profile_all_configs.sh [tag_prefix]
cd $GRAFT_REPO_ROOT
P=${1:-r03}
for cfg in "room|--scene 3d_room.json --max-depth 8" "hallways|--scene 3d_hallways.json --max-depth 12" "4dframe|--scene 4d_frame.json --max-depth 8"; do
  tag=${cfg%%|*}; args=${cfg#*|}
  BENCH_ARGS="$args" bash tools/kernel_stats.sh ${P}_$tag > gpurun_out/${P}_${tag}_kernel_stats.txt 2>&1
  BENCH_ARGS="$args --streams 1 --frames-in-flight 1" bash tools/kernel_stats.sh ${P}_${tag}_1s > gpurun_out/${P}_${tag}_kernel_stats_single_stream.txt 2>&1
  BENCH_ARGS="$args --streams 1 --frames-in-flight 1 --no-other-configs" bash tools/frame_dispatches.sh ${P}_$tag > gpurun_out/${P}_${tag}_dispatches.txt 2>&1
  BENCH_ARGS="$args" PMC_DIR=pmc_$tag bash tools/pmc_passes.sh > gpurun_out/${P}_${tag}_pmc_passes.log 2>&1
  echo "done $tag"
done

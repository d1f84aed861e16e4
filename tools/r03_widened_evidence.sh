#!/bin/bash
# round 3: kernel stats of the widened rows on the final build: LIIF, MetaSR, decoder training
# step, whole DIINN forward; plus their plain timings against the eager op sequences.
#   gpurun --timeout 1500 -- 'bash tools/r03_widened_evidence.sh'
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r03w; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python3 $R/tools/liif_time.py 256 4 > $O/liif_time.txt 2>&1
python3 $R/tools/metasr_time.py 256 4 > $O/metasr_time.txt 2>&1
python3 $R/tools/train_time.py 16 48 4 > $O/train_time.txt 2>&1
python3 $R/tools/e2e_time.py > $O/e2e_times.txt 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/liif -- python3 $R/tools/liif_time.py 256 4 --only-ours > $O/liif_prof.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/metasr -- python3 $R/tools/metasr_time.py 256 4 --only-ours > $O/metasr_prof.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/train -- python3 $R/tools/train_time.py 16 48 4 --only-ours > $O/train_prof.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/e2e -- python3 $R/tools/e2e_profile.py 256 4 10 > $O/e2e_prof.log 2>&1
cd $R
python tools/summarize_rocprof.py $O/liif $O/r03_liif_kernel_stats.csv "tools/liif_time.py 256 4 --only-ours (LIIF decode, 256x256 -> 1024x1024)" > /dev/null 2>&1
python tools/summarize_rocprof.py $O/metasr $O/r03_metasr_kernel_stats.csv "tools/metasr_time.py 256 4 --only-ours (MetaSR decode, 256x256 -> 1024x1024)" > /dev/null 2>&1
python tools/summarize_rocprof.py $O/train $O/r03_train_kernel_stats.csv "tools/train_time.py 16 48 4 --only-ours (decoder forward + backward, B=16, 48x48 -> 192x192)" > /dev/null 2>&1
python tools/summarize_rocprof.py $O/e2e $O/r03_e2e_256_kernel_stats.csv "tools/e2e_profile.py 256 4 10 (whole DIINN forward, 10 calls)" > /dev/null 2>&1
find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete
tail -4 $O/liif_time.txt $O/metasr_time.txt $O/train_time.txt; head -8 $O/r03_liif_kernel_stats.csv | cut -c1-140

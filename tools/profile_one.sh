#!/bin/bash
# rocprofv3 kernel-trace stats of one bench configuration:   bash tools/profile_one.sh <tag> <name> [bench args...]
# raw output under gpurun_out/prof/<name> (removed first: rocprofv3 -d accumulates one subdirectory per run), summary copied
# to profiles/<tag>_kernel_stats_<name>.csv and gpurun_out/profiles_<tag>/ (profiles/ itself does not travel back from the box)
set -o pipefail
TAG=$1; NAME=$2; shift 2
OUT=gpurun_out/prof
mkdir -p $OUT profiles gpurun_out/profiles_$TAG
export TMPDIR=/tmp
rm -rf $OUT/$NAME
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$NAME -- python3 bench.py --other-encoders none --no-trace --no-cpu-baseline --no-parity --no-e2e "$@" > $OUT/$NAME.json 2> $OUT/$NAME.err || exit 1
f=$(ls -t $OUT/$NAME/*/*kernel_stats.csv | head -1)
cp "$f" profiles/${TAG}_kernel_stats_$NAME.csv
cp $OUT/$NAME.json profiles/${TAG}_bench_under_rocprof_$NAME.json
cp profiles/${TAG}_kernel_stats_$NAME.csv profiles/${TAG}_bench_under_rocprof_$NAME.json gpurun_out/profiles_$TAG/
echo "== $NAME: $(python3 -c "import json;d=json.load(open('$OUT/$NAME.json'));print(d['value'], d['unit'], d['config']['ms_per_batch'], 'ms/batch', d.get('verified'))")"
head -12 "$f" | cut -c1-160

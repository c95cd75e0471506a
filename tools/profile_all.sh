#!/bin/bash
# Round profile set, run on the GPU box from the repo root:   bash tools/profile_all.sh r05 [all|stats|pmc|traffic]
# Kernel-trace stats of the TIMED path (hipGraph replay of two concurrent utterance groups; the two warm-up forwards
# before capture are < 2 % of the launches) for every speech config, then the PMC passes on the headline config
# (separate passes for FETCH_SIZE / WRITE_SIZE / SQ counters, --kernel-trace only, as the MI355X guide prescribes).
# Raw output under gpurun_out/prof/, summaries under profiles/<tag>_*.
set -o pipefail
TAG=${1:-r05}
OUT=gpurun_out/prof
mkdir -p $OUT profiles
cd /tmp 2>/dev/null; export TMPDIR=/tmp; cd - >/dev/null
FLAGS="--no-trace --no-cpu-baseline --no-parity --no-e2e --other-encoders none"
stats() {  # name, bench args...
    local name=$1; shift
    rm -rf $OUT/$name                       # rocprofv3 -d accumulates one subdirectory per run: never stamp a stale CSV
    rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$name -- python3 bench.py --other-encoders none $FLAGS "$@" > $OUT/$name.json 2> $OUT/$name.err || return 1
    local f=$(ls -t $OUT/$name/*/*kernel_stats.csv | head -1)
    cp "$f" profiles/${TAG}_kernel_stats_$name.csv
    cp $OUT/$name.json profiles/${TAG}_bench_under_rocprof_$name.json
    echo "== $name: $(python3 -c "import json;d=json.load(open('$OUT/$name.json'));print(d['value'], d['unit'], d['config']['ms_per_batch'], 'ms/batch', d.get('verified'))")"
    head -8 "$f" | cut -c1-150
}
rm -rf $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_sq $OUT/pmc_l2 $OUT/wpmc_fetch $OUT/wpmc_write
PMC="--steps 2 --warmup 1 --reps 1 --no-graph --no-verify $FLAGS"
if [ "$2" = "traffic" ]; then   # only the two traffic passes: re-stamp profiles/<tag>_pmc_traffic.json after a kernel-source change
    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- python3 bench.py --other-encoders none $PMC > /dev/null 2> $OUT/pmc_fetch.err || exit 1
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- python3 bench.py --other-encoders none $PMC > /dev/null 2> $OUT/pmc_write.err || exit 1
    python3 tools/pmc_summary.py $OUT/pmc_fetch $OUT/pmc_write profiles/$TAG "microsoft/wavlm-large|bf16|batch=16x10s|inflight=2|groups=1"
    mkdir -p gpurun_out/profiles_$TAG && cp profiles/${TAG}_pmc_* gpurun_out/profiles_$TAG/
    echo done; exit 0
fi
PART=${2:-all}       # all | stats | pmc | traffic | default (= the default mode's kernel stats only)  (stats and pmc as two gpurun calls when one call's limit is too short for both)
if [ "$PART" = "default" ]; then
    stats wavlm_large_f16mf --steps 10 --mode f16mf || exit 1
    mkdir -p gpurun_out/profiles_$TAG && cp profiles/${TAG}_*f16mf* gpurun_out/profiles_$TAG/; echo done; exit 0
fi
if [ "$PART" != "pmc" ]; then
stats wavlm_large_bf16 --steps 10 || exit 1
stats wavlm_large_f16x --steps 10 --mode f16x || exit 1
stats wavlm_large_f16mf --steps 10 --mode f16mf || exit 1     # the drivers' default (round 5)
stats wavlm_large_f16m --steps 10 --mode f16m || exit 1
stats wavlm_large_f16a --steps 10 --mode f16a || exit 1
# one launch at a time (no graph, one batch, no concurrent branch): the CSV from which roofline.one_launch_at_a_time.avg_launch_us can be
# reproduced (the other CSVs' averages are inflated by the two branches overlapping under the profiler)
stats wavlm_large_bf16_one_launch_at_a_time --steps 3 --no-graph --inflight 1 --micro 1 || exit 1
cp profiles/${TAG}_kernel_stats_wavlm_large_bf16_one_launch_at_a_time.csv profiles/${TAG}_kernel_trace_one_launch_at_a_time_wavlm_large_bf16.csv
stats hubert_xlarge_bf16 --steps 5 --ssl_type facebook/hubert-xlarge-ll60k || exit 1
stats xlsr_2b_bf16 --steps 5 --ssl_type facebook/wav2vec2-xls-r-2b --batch 8 || exit 1
stats whisper_large_v3_bf16 --steps 3 --reps 4 --ssl_type openai/whisper-large-v3 --seconds 30 || exit 1
fi
if [ "$PART" = "stats" ]; then mkdir -p gpurun_out/profiles_$TAG && cp profiles/${TAG}_* gpurun_out/profiles_$TAG/; echo done; exit 0; fi
# PMC passes: eager launches (counters are per dispatch), headline workload
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- python3 bench.py --other-encoders none $PMC > /dev/null 2> $OUT/pmc_fetch.err || exit 1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- python3 bench.py --other-encoders none $PMC > /dev/null 2> $OUT/pmc_write.err || exit 1
python3 tools/pmc_summary.py $OUT/pmc_fetch $OUT/pmc_write profiles/$TAG "microsoft/wavlm-large|bf16|batch=16x10s|inflight=2|groups=1"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE \
    --kernel-trace --output-format csv -d $OUT/pmc_sq -- python3 bench.py --other-encoders none $PMC > /dev/null 2> $OUT/pmc_sq.err || exit 1
# second SQ pass: what the K loop waits on (LDS issue stalls, vector and matrix instructions executing together)
rm -rf $OUT/pmc_sq2
rocprofv3 --pmc SQ_VALU_MFMA_COEXEC_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE \
    --kernel-trace --output-format csv -d $OUT/pmc_sq2 -- python3 bench.py --other-encoders none $PMC > /dev/null 2> $OUT/pmc_sq2.err || echo "(second SQ pass not available: $(tail -1 $OUT/pmc_sq2.err))"
python3 tools/pmc_sq_summary.py $OUT/pmc_sq profiles/$TAG $OUT/pmc_sq2
# the f16m step: matrix-pipe busy share and the instruction mix of its GEMMs (fp16 MFMAs next to the block-scaled e4m3 ones)
rm -rf $OUT/pmc_sq_f16m $OUT/pmc_sq2_f16m
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE \
    --kernel-trace --output-format csv -d $OUT/pmc_sq_f16m -- python3 bench.py --other-encoders none $PMC --mode f16m > /dev/null 2> $OUT/pmc_sq_f16m.err || exit 1
rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_VALU_MFMA_MOPS_F8 SQ_INSTS_VALU_MFMA_F16 SQ_INSTS_VALU_MFMA_F8 SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE \
    --kernel-trace --output-format csv -d $OUT/pmc_sq2_f16m -- python3 bench.py --other-encoders none $PMC --mode f16m > /dev/null 2> $OUT/pmc_sq2_f16m.err || echo "(f16m instruction-mix pass not available: $(tail -1 $OUT/pmc_sq2_f16m.err))"
python3 tools/pmc_sq_summary.py $OUT/pmc_sq_f16m profiles/${TAG}_f16m $OUT/pmc_sq2_f16m
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $OUT/pmc_l2 -- python3 bench.py --other-encoders none $PMC > /dev/null 2> $OUT/pmc_l2.err || exit 1
python3 tools/pmc_l2_summary.py $OUT/pmc_l2 profiles/$TAG
# Whisper front end alone against HBM bytes (FETCH/WRITE of the logmel kernels come out of the per-kernel CSV)
WPMC="--steps 1 --warmup 1 --reps 1 --no-graph --no-verify $FLAGS --ssl_type openai/whisper-large-v3 --seconds 30"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/wpmc_fetch -- python3 bench.py --other-encoders none $WPMC > /dev/null 2> $OUT/wpmc_fetch.err || exit 1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/wpmc_write -- python3 bench.py --other-encoders none $WPMC > /dev/null 2> $OUT/wpmc_write.err || exit 1
python3 tools/pmc_summary.py $OUT/wpmc_fetch $OUT/wpmc_write profiles/${TAG}_whisper "openai/whisper-large-v3|bf16|batch=16x30s|inflight=1|groups=2"
mkdir -p gpurun_out/profiles_$TAG && cp profiles/${TAG}_* gpurun_out/profiles_$TAG/      # profiles/ itself does not travel back
echo done

run() { python bench.py --other-encoders none --no-cpu-baseline --no-trace "$@" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['metric'][:70], '|', d['value'], d['unit'], d['ms_per_step'], 'ms', d.get('achieved_tflops_whole_path'))"; }
run
run --mode fp32x
run --ssl_type facebook/hubert-xlarge-ll60k
run --ssl_type facebook/wav2vec2-xls-r-2b --batch 8
run --ssl_type openai/whisper-large-v3 --seconds 30
run --ssl_type roberta-large --batch 64
run --ssl_type microsoft/deberta-v3-large --batch 64

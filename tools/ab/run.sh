for i in 1 2 3; do
for v in old new; do
SER_HIP_LIB=$PWD/tools/ab/lib_$v.so python bench.py --no-cpu-baseline --no-trace 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', d['value'], d['ms_per_step'])"
done; done

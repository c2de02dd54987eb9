export TMPDIR=/tmp
for F in "" "--swin-graph"; do
  python bench.py --arch swinv2t --steps 30 --warmup 5 --no-cpu-baseline $F 2>/tmp/err.txt | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('swin $F', d['value'], d['ms_per_step'], 'host', d['host_ms_per_step'], 'blocked', d['host_blocked_on_rings_ms_per_step'], d['config']['launch'][:20], d['config']['loss'])"
  grep "\[bench\]" /tmp/err.txt | head -3
done

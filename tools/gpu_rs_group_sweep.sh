pr() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['ms_per_step'], d.get('kernel_ms_per_step'))"; }
for wl in c2 c4; do
ST=300; [ $wl = c4 ] && ST=80
for g in 4 8 16 32; do for m in 2048 6144 12288; do
 HT_DEBUG_RS_GROUP=$g HT_DEBUG_RS_MINWG=$m python bench.py --workload $wl --steps $ST --cpu-seconds 0 --no-sub 2>/dev/null | pr "group=$g minwg=$m $wl"
done; done; done

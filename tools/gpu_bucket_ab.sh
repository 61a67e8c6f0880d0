show() { tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('value %.2f G/s  fused/step %.4f ms'%(d['value']/1e9, r['avg_launch_ms']/r['steps_per_launch']))"; }
for mode in exact fast; do
for b in 0 1 0 1; do echo -n "$mode TH_BUCKET=$b: "; TH_BUCKET=$b python bench.py --no-cpu --no-traffic --mode $mode 2>&1 | show; done
echo -n "$mode in-view auto: "; python bench.py --no-cpu --no-traffic --mode $mode --in-view 2>&1 | show
echo -n "$mode in-view TH_BUCKET=0: "; TH_BUCKET=0 python bench.py --no-cpu --no-traffic --mode $mode --in-view 2>&1 | show
done

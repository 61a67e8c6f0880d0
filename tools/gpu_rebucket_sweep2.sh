show() { tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('value %.2f G/s  fused/step %.4f ms'%(d['value']/1e9, r['avg_launch_ms']/r['steps_per_launch']))"; }
for r in 1 2; do
echo -n "TH_BUCKET=0: "; TH_BUCKET=0 python bench.py --no-cpu --no-traffic 2>&1 | show
for iv in 128 256 512; do echo -n "auto every $iv: "; TH_REBUCKET_STEPS=$iv python bench.py --no-cpu --no-traffic 2>&1 | show; done
done

# one-box A/B of the fp64 down-date at C5 (50k landmarks, Joseph form): the product library against other builds (arguments)
run() { label=$1; shift; env "$@" 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print('$label syrk_ms', round(j['roofline']['avg_launch_ms'],4), 'step_ms', round(j['ms_per_step'],4), 'frac', round(j['roofline']['frac'],3))
"; }
B="timeout -k 10 300 python bench.py --no-cpu-baseline --no-fastslam --no-pmc --steps 20 --warmup 3 --landmarks 50000 --obs 8 --dtype f64 --form joseph"
for rep in 1 2; do
run "C5 product" $B
for other in "$@"; do run "C5 $other" SLAMHIP_LIBRARY=$other $B; done
done

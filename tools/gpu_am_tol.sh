for t in 0 2 5 20 50; do
  FMR_X_AMTOL=$t timeout 300 python bench.py --mode am --steps 20 --warmup 3 --no-cpu-baseline < /dev/null 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        b=json.loads(l); print('amtol', $t, b['value'], b['ms_per_step'], b['recurrences']['agc_newton_rounds'], b['recurrences']['agc_residuals'], b['audio_check'])
"
done

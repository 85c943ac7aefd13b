for i in 1 2 3; do
  for v in 0 1; do
    DRN_TUNE=12=$v python bench.py --steps 300 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.readline()); print('pp=$v', round(d['value'],1), 'img/s', round(d['ms_per_step'],4), 'ms; dominant in-step TF', round(d['roofline_step']['dominant_kernel_tflops_in_step'],1), 'fc6 fwd', round(d['roofline']['achieved'],1))"
  done
done

for m in 1 3 1 3; do
python -c "
import sys, runpy
from fami_pose_amd._lib import lib
lib().cdll.fami_conv_tune_xcd($m)
sys.argv = ['bench.py', '--steps', '10', '--warmup', '3', '--no-cpu-baseline', '--also', 'bf16']
runpy.run_path('bench.py', run_name='__main__')
" 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('xcd mode $m:', d['ms_per_step'], d['also_bf16']['ms_per_step'])"
done

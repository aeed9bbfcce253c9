import sys, os
sys.path.insert(0, os.getcwd())
import pytest
from fami_pose_amd._lib import lib
L = lib().cdll
for knobs in sys.argv[1:]:
    for k in knobs.split(','):
        L.fami_conv_tune_stages(int(k))
    print('=== knobs', knobs, flush=True)
    pytest.main(['tests/test_model_gpu.py', '-q', '-x', '-k', 'test_baseline_configs_4_5_forward and 32-2', '--no-header', '-p', 'no:cacheprovider'])
    L.fami_conv_tune_stages(101); L.fami_conv_tune_stages(111)

python -m pytest tests/test_gpu_models.py -x -q -k "boost" 2>&1 | tail -15
python - <<'PY'
import sys, time, torch, numpy as np
sys.path.insert(0, 'stable-diffusion-webui-depthmap-script_amd')
from src import boost
from lib.multi_depth_model_woauxi import RelDepthModel
from pix2pix.models.pix2pix4depth_model import Pix2Pix4DepthModel
torch.manual_seed(0)
net = RelDepthModel('resnext101').eval().cuda(); p2p = Pix2Pix4DepthModel().eval().cuda()
rng = np.random.default_rng(1)
yy, xx = np.mgrid[0:2160, 0:3840]
img = (127 + 60*np.sin(xx/37.0)[...,None]*np.cos(yy/23.0)[...,None] + rng.normal(0,25,(2160,3840,3))).clip(0,255).astype(np.uint8)
t = torch.from_numpy(img).cuda()
for i in range(2):
    st = {}; torch.cuda.synchronize(); t0 = time.perf_counter()
    out = boost.estimateboost(t, net, 0, p2p, 1600, stats=st); torch.cuda.synchronize()
    print('4K boost run', i, '%.2f s' % (time.perf_counter()-t0), st)
PY

python - <<'PY'
import time, sys
sys.path.insert(0,'.'); sys.path.insert(0,'tests')
t=time.time(); import torch; print('import torch', round(time.time()-t,1))
from conftest import load_golden
from s3prl_amd.encoder import HipEncoder
for name in ("hubert_large_s1_pl","hubert_base_s1_pl"):
    t=time.time(); meta,cfg,w,wavs,g,_=load_golden(name); print(name,'load_golden (synth weights)', round(time.time()-t,2))
    dev=[torch.from_numpy(x).cuda() for x in wavs]
    for mode in ("fp32","fp32x3","fp16x2","fp16","bf16"):
        t=time.time(); enc=HipEncoder(cfg,w,dtype=mode); t1=time.time()-t
        t=time.time(); hs=enc.forward(dev).cpu(); t2=time.time()-t
        enc.close(); print('  ',mode,'create',round(t1,2),'forward+copy',round(t2,2))
PY

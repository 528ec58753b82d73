"""Host-side cost of one forward (time for the asynchronous s3enc_forward call to return on an empty queue) next to the
steady-state GPU time per forward."""
import sys, time, torch
sys.path.insert(0, "/root/repo")
from s3prl_amd.encoder import HipEncoder
from s3prl_amd.synth import named_config, synth_weights
for model, dtype in (("hubert_base", "bf16"), ("hubert_large", "bf16")):
    cfg = named_config(model); enc = HipEncoder(cfg, synth_weights(cfg, 0), dtype=dtype)
    wavs = [torch.randn(160000, device="cuda") for _ in range(32)]
    out = enc.forward(wavs); torch.cuda.synchronize()
    sub = []
    for _ in range(5):  # one forward at a time on an empty queue: pure host-side submission cost
        torch.cuda.synchronize(); t0 = time.perf_counter(); enc.forward(wavs, out=out); sub.append(time.perf_counter() - t0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10): enc.forward(wavs, out=out)
    torch.cuda.synchronize(); t2 = time.perf_counter()
    print(model, dtype, "host submit ms/forward (empty queue):", min(sub) * 1e3, "steady-state ms/forward:", (t2 - t0) / 10 * 1e3)
    enc.close()

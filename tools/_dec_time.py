import sys, os, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests", "golden"))
import torch
from gags_amd.decoders import CNN_decoder, CNN_scale_decoder
dec, sdec = CNN_decoder(16, 512).cuda(), CNN_scale_decoder(16, 3).cuda()
x = torch.randn(1080, 1920, 16, device="cuda").permute(2, 0, 1)
with torch.no_grad():
    for _ in range(2): y = dec(x); s = sdec(x)
    torch.cuda.synchronize()
    for name, f in (("CNN_decoder", dec), ("CNN_scale_decoder", sdec)):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): f(x)
        e1.record(); torch.cuda.synchronize()
        print(name, "fwd 1080p ms:", e0.elapsed_time(e1) / 5)

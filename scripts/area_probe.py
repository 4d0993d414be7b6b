import sys, os, torch, numpy as np
sys.path.insert(0, os.getcwd())
import livevisionkit_amd as lvk
ws = torch.cuda.Stream(); ctx = lvk.Context(0, stream=ws)
for shape in [(1440, 2560), (1200, 1920), (720, 1280), (2160, 3840), (1080, 1920)]:
    for packed in (False, True):
        src = torch.randint(0, 256, shape + ((3,) if packed else ()), dtype=torch.uint8, device="cuda")
        with torch.cuda.stream(ws):
            for _ in range(20): ctx.luma_area_resize(src, 270, 480)
            a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
            a.record(ws)
            for _ in range(200): ctx.luma_area_resize(src, 270, 480)
            b.record(ws)
        ctx.sync(); torch.cuda.synchronize()
        print(shape, "packed" if packed else "planar", "%.2f us per call (back to back)" % (a.elapsed_time(b) * 1e3 / 200))

import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from deepinpainting_amd import ops
B = 16
g = torch.Generator(device="cuda").manual_seed(1)
for tr, Cin, Cout, S, k, st in ((False, 6, 64, 256, 3, 1), (True, 128, 3, 256, 3, 1), (False, 3, 64, 256, 4, 2)):
    x = (torch.rand(B, Cin, S, S, device="cuda", generator=g) * 2 - 1).to(torch.bfloat16)
    So = (S - 1) * st - 2 + k if tr else (S + 2 - k) // st + 1
    dy = (torch.rand(B, Cout, So, So, device="cuda", generator=g) * 2 - 1).to(torch.bfloat16)
    for _ in range(5):
        ops.conv_thin_wrw_mfma(tr, x, dy, k, st)
torch.cuda.synchronize()

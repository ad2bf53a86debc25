import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from llm_awq_amd import ops
g = np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "repack_v1_v2.npz"))
for i in range(2):
    dt = torch.float16 if int(g[f"dtype_{i}"][0]) == 0 else torch.bfloat16
    qw1 = torch.from_numpy(g[f"qw1_{i}"]).cuda(); qz1 = torch.from_numpy(g[f"qz1_{i}"]).cuda()
    sc1 = torch.from_numpy(g[f"sc1_{i}"]).view(dt).cuda()
    qw2, s2, sz2 = ops.repack_v1_to_v2(qw1, sc1, qz1)
    got = sz2.cpu().view(torch.int16).numpy(); ref = g[f"sz2_{i}"]
    bad = np.argwhere(got != ref)
    print(i, dt, "mismatch", len(bad), "of", got.size, "shape", got.shape)
    for (a, b) in bad[:8]:
        print("   at", a, b, "got", hex(int(got[a, b]) & 0xFFFF), "ref", hex(int(ref[a, b]) & 0xFFFF),
              "scale bits", hex(int(g[f'sc1_{i}'][b, a]) & 0xFFFF))

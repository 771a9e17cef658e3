"""Bit-identity of the scorer across two builds of the library (TEXTGAN_HIP_LIB): run once per build, compare the files.

    TEXTGAN_HIP_LIB=<so> python tools/diag/conv_bits.py <out.npz>
Raw rewards, features and logits at configs[1]'s shape (rollout prefix mode and plain mode), SentiGAN's table, L = 20."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import textgan_pytorch_amd as tg  # noqa: E402

dev = "cuda"
out = {}
for name, kind, V, L, B, R in (("c2", "seqgan", 5255, 51, 64, 4), ("c1", "seqgan", 5000, 20, 64, 4), ("senti", "sentigan", 5000, 20, 30, 3)):
    dis = tg.modules.synthetic_discriminator(kind, V, seed=0, k_label=4, device=dev).eval()
    sent = torch.from_numpy(tg.synth.tokens(1, (B, L), V)).to(dev)
    comp = torch.from_numpy(tg.synth.tokens(2, (R, L, B, L), V)).to(dev)
    for g in range(1, L + 1):                     # completions share their first g tokens with the sentence
        comp[:, g - 1, :, :g] = sent[:, :g]
    raw = tg.ops.dis_score_rollout(dis, sent, comp, col=1)
    r = tg.ops.dis_score(dis, comp.view(-1, L)[:1000], want_logits=True, want_feature=True)
    out[name + "_raw"] = raw.cpu().numpy()
    out[name + "_feature"] = r["feature"].cpu().numpy()
    out[name + "_logits"] = r["logits"].cpu().numpy()
np.savez(sys.argv[1], **out)
print("wrote", sys.argv[1], tg._lib.LIB_PATH)

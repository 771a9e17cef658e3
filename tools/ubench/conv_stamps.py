# Per-wave phase cycles of k_dis_conv<Prefix, split> at C2 (diagnostic build path: TG_CONV_STAMPS).
import os, sys, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import textgan_pytorch_amd as tg
V, L, B, R = 5255, 51, 64, 16
dev = "cuda"
gen = tg.modules.synthetic_generator(V, 32, 32, L, device=dev)
dis = tg.modules.synthetic_discriminator("seqgan", V, seed=0, device=dev).eval()
sent = torch.from_numpy(tg.synth.tokens(1, (B, L), V)).to(dev)
u = torch.rand(R, L, L, B, device=dev)
comp = tg.ops.gen_rollout(gen, sent, R, u)
tg.ops.dis_score_rollout(dis, sent, comp)
nwg = R * L * B // 8
st = torch.zeros(nwg * 8 * 8 + 1024, dtype=torch.int64, device=dev)
os.environ["TG_CONV_STAMPS"] = str(st.data_ptr())
tg.ops.dis_score_rollout(dis, sent, comp); torch.cuda.synchronize()
del os.environ["TG_CONV_STAMPS"]
s = st[:nwg * 64].cpu().numpy().astype(np.float64).reshape(nwg, 8, 8)
gather, loop, mfma, items = s[..., 0], s[..., 1], s[..., 2], s[..., 3]
wg_total = (gather + loop).max(1)
print("workgroups", nwg, "mean WG cycles %.0f" % wg_total.mean())
print("gather  mean %.0f (%.1f%% of WG)" % (gather.mean(), 100 * gather.mean() / wg_total.mean()))
print("item loop per wave mean %.0f, in-item (MFMA stages + chunk epilogues) %.0f (%.1f%% of WG), items/wave %.1f"
      % (loop.mean(), mfma.mean(), 100 * mfma.mean() / wg_total.mean(), items.mean()))
print("item fetch (counter, descriptor) per wave mean %.0f (%.1f%%), item epilogue %.0f (%.1f%%)"
      % (s[..., 4].mean(), 100 * s[..., 4].mean() / wg_total.mean(), s[..., 5].mean(), 100 * s[..., 5].mean() / wg_total.mean()))
print("wave idle at the end of the WG: mean %.0f (%.1f%%)" % ((wg_total[:, None] - gather - loop).mean(),
      100 * (wg_total[:, None] - gather - loop).mean() / wg_total.mean()))
# workgroup id -> given_num under the per-XCD order of k_dis_conv<Prefix> (8 | B/S): slot = id >> 3 = (g, j, i), i fastest
nb = B // 8
g = (np.arange(nwg) >> 3) // (R * (nb // 8)) + 1
for lo, hi in ((1, 5), (20, 30), (45, 50)):
    m = (g >= lo) & (g <= hi)
    print("g in [%d,%d]: WG cycles %.0f gather %.0f in-item %.0f" % (lo, hi, wg_total[m].mean(), gather[m].mean(), mfma[m].mean()))

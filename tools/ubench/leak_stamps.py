# Phase cycles of k_leak_cells (diagnostic build: make DIAG=1) at run_leakgan.py's shape.
import os, sys, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import textgan_pytorch_amd as tg
V, E, H, K, L, B, R, step = 5000, 32, 32, 16, 20, 64, 4, 4
dev = "cuda"
gen = tg.modules.synthetic_leakgan(V, E, H, L, goal_size=K, step_size=step, batch_size=B, device=dev)
dis = tg.modules.synthetic_discriminator("seqgan", V, seed=0, device=dev)
sent = torch.from_numpy(tg.synth.tokens(1, (B, L), V)).to(dev)
ro = tg.ROLLOUT(gen)
ro.get_reward_leakgan(sent, R, dis, 0)
nwg = (R * (L // step) * B + 31) // 32
st = torch.zeros(nwg * 8 + 64, dtype=torch.int64, device=dev)
os.environ["TG_LEAK_STAMPS"] = str(st.data_ptr())
ro.get_reward_leakgan(sent, R, dis, 0); torch.cuda.synchronize()
del os.environ["TG_LEAK_STAMPS"]
s = st[:nwg * 8].cpu().numpy().astype(np.float64).reshape(nwg, 8)
n = s[:, 4].mean()
print("launches per WG", n)
for i, name in enumerate(("worker cell", "z operand", "manager cell", "goal algebra")):
    print("%-14s %8.0f cycles per launch" % (name, (s[:, i] / s[:, 4]).mean()))

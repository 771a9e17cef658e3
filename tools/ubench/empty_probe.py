import sys, torch, numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import textgan_pytorch_amd as tg
V, L = 50, 20
dev = "cuda"
gen = tg.modules.synthetic_generator(V, 32, 32, L, device=dev)
dis = tg.modules.synthetic_discriminator("seqgan", V, seed=0, device=dev).eval()
def tryit(name, f):
    try:
        r = f()
        print(name, "->", {k: tuple(v.shape) for k, v in r.items()} if isinstance(r, dict) else (tuple(r.shape) if hasattr(r, "shape") else r))
    except Exception as e:
        print(name, "RAISES", type(e).__name__, str(e)[:150])
e = torch.zeros(0, L, dtype=torch.long, device=dev)
tryit("dis_score empty", lambda: tg.ops.dis_score(dis, e, want_logits=True, want_feature=True))
tryit("dis.forward empty", lambda: tg.discriminator.forward(dis, e))
tryit("get_reward B=0", lambda: tg.ROLLOUT(gen).get_reward(e, 2, dis))
tryit("get_reward R=0", lambda: tg.ROLLOUT(gen).get_reward(torch.zeros(3, L, dtype=torch.long, device=dev), 0, dis))
tryit("teacher empty", lambda: tg.ops.gen_teacher_forced(gen, e, e))
tryit("nll empty", lambda: tg.ops.gen_nll(gen, e, e))
tryit("sample 0", lambda: tg.generator.sample(gen, 0, 4, 1))
tryit("pg empty", lambda: tg.pgloss.batchPGLoss(gen, e, e, torch.zeros(0, L, device=dev)))
tryit("mali B=0", lambda: tg.maligan.get_mali_reward(dis, e, 2))

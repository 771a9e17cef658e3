"""Disagreements of the two generator paths with a golden's completions, with the CDF gap at each (diagnostic)."""
import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import _cases
import textgan_pytorch_amd as tg
from oracle import oracle

name = sys.argv[1] if len(sys.argv) > 1 else "reward_v20k_h64"
c = _cases.load(name)
gen = tg.modules.load_numpy_state(tg.modules.LSTMGenerator(int(c["E"]), int(c["H"]), int(c["V"]), int(c["L"])), _cases.gen_sd(c), "cuda")
sent, u, _ = _cases.reward_inputs(c)
R = int(c["R"])
ref = c["completions"].astype(np.int64)
og = oracle.Gen(_cases.gen_sd(c))
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
m = _cases.sampled_mask(*ref.shape[:3])
for prec in ("fp32", "split_f16"):
    rep = tg.ops.gen_rollout(gen, t(sent), R, t(u), forced=t(ref), precision=prec).cpu().numpy()
    diff = (rep != ref) & m
    print(prec, "disagreements", int(diff.sum()), "of", int(m.sum()))
    for (i, gi, b, tt) in np.argwhere(diff):
        row = ref[i, gi, b]
        r = oracle.gen_teacher_forced(og, row[None, :], want_logits=True)
        lg = r["logits"][0, tt - 1].astype(np.float64)
        w = np.exp(lg - lg.max()); cdf = np.cumsum(w) / w.sum()
        a, cc = int(rep[i, gi, b, tt]), int(ref[i, gi, b, tt])
        uu = float(u[i, gi, tt, b])
        lo, hi = min(a, cc), max(a, cc)
        print("  at", (i, gi + 1, b, tt), "got", a, "ref", cc, "u", uu, "cdf gaps", (cdf[lo:hi] - uu)[:4], "p", w[lo] / w.sum())

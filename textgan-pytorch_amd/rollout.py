"""Drop-in for the reference's ``utils.rollout.ROLLOUT`` (utils/rollout.py:15-154).

Same constructor, same ``get_reward(sentences, rollout_num, dis, current_k=0)``
signature and result ([B, L] float tensor on the input's device), same
attributes.  The double Python loop over (rollout, given_num), the per-token
``gen.forward`` / ``torch.multinomial`` launches and the per-prefix
``dis.forward`` calls collapse into three HIP launches (prefix states, all
completions, scorer) plus one RCCL all-gather when a process group is active.
"""
import copy

import torch

from . import _lib, dist as tdist, ops


class ROLLOUT:
    #: sequences scored per tg_dis_score call (bounds the pooled / dropout scratch)
    dis_chunk = 1 << 17
    #: conv arithmetic of the scorer: None = library default ("split_f16"), or "fp32" (see TG_DIS_*)
    dis_precision = None
    #: arithmetic of the generator's vocabulary projection: None = library default, "fp32" or "split_f16" (TG_GEN_*)
    gen_precision = None
    #: dropout masks of the scorer when no dropout_uniforms are passed: "kernel" = counter-based, generated inside the
    #: highway epilogue from a per-call seed (no [N, F] tensor); "torch" = torch.rand uniforms (ROLLOUT.generator)
    dropout_rng = "kernel"
    #: with a process group: broadcast weights from rank 0 of the group at the top of every get_reward call (once per
    #: adversarial step).  True / "both": the generator's and the scorer's; "gen" / "dis": that module
    #: only (SentiGAN's k generators share one D that nobody steps between their calls: the first call of a step
    #: carries it, the others say "gen"); False: none -- the trainer keeps the replicas equal by other means (DDP, an
    #: explicit tg.dist.sync_weights).  The value must be the same on every rank of the group: it decides whether a
    #: collective is entered.
    sync_weights = True
    #: get_reward enqueues the scorer's preparation (what needs only D's parameters and the sentences: weight pieces,
    #: the sentences' window values) on a second stream, beside the generator's kernels.  Same results to the bit;
    #: False keeps the whole call on the current stream.
    overlap_prepare = True

    def __init__(self, gen, gpu=True):
        self.gen = gen
        self._old_model = None          # reference deep-copies here (rollout.py:18) and never reads it
        self.max_seq_len = gen.max_seq_len
        self.vocab_size = gen.vocab_size
        self.step_size = gen.step_size if getattr(gen, "name", None) == "leakgan" else 0
        self.goal_out_size = gen.goal_out_size if getattr(gen, "name", None) == "leakgan" else 0
        self.gpu = gpu
        self.generator = None           # optional torch.Generator for the uniforms
        self.group = None               # sharding is opt-in: tg.dist.WORLD (default group) or a ProcessGroup
        self._rank_gens = {}            # (device, rank) -> torch.Generator of a sharded call's private uniforms
        self.timers = None              # optional dict name -> list[(start_event, end_event)]
        self._side = {}                 # device -> torch.cuda.Stream of the scorer's preparation
        _lib.lib()                      # fail loudly here if the HIP extension is missing

    @property
    def old_model(self):
        if self._old_model is None:
            self._old_model = copy.deepcopy(self.gen)
        return self._old_model

    # ------------------------------------------------------------------ helpers
    def _timed(self, name):
        return _Timed(self.timers, name)

    def _check(self, sentences):
        if not sentences.is_cuda:
            raise RuntimeError("textgan_hip ROLLOUT: sentences must be on the GPU (got %s); there is no CPU path"
                               % sentences.device)
        if sentences.dim() != 2 or sentences.shape[1] != self.max_seq_len:
            raise RuntimeError("textgan_hip ROLLOUT: sentences must be [batch, %d]" % self.max_seq_len)

    def _sync(self, dis):
        sel = self.sync_weights
        if sel is True:
            sel = "both"
        if sel not in ("both", "gen", "dis"):
            raise RuntimeError("textgan_hip ROLLOUT: sync_weights must be True, False, 'both', 'gen' or 'dis', got %r"
                               % (self.sync_weights,))
        with self._timed("sync_weights"):
            tdist.sync_weights(self.gen if sel in ("both", "gen") else None,
                               dis if sel in ("both", "dis") else None, self.group)

    def _rand(self, shape, device, plan=None):
        """U[0,1) draws.  Single rank: ROLLOUT.generator (None = torch's default).  Sharded: every rank holds the same
        torch seed, so each draws from a private generator seeded from (base seed, rank) -- otherwise rank k's
        completions would be bit-copies of rank 0's and the effective rollout_num R / world.  The base seed is
        ROLLOUT.generator's initial_seed() (torch.initial_seed() without one); the private generator is rebuilt
        whenever that seed or the generator object changes, so torch.manual_seed / generator.manual_seed re-seed a
        sharded run as they re-seed a single-rank one (the state of the user's generator itself is not consumed)."""
        gen = self.generator
        if plan is not None and plan.world > 1:
            base = self.generator.initial_seed() if self.generator is not None else torch.initial_seed()
            key = (str(device), plan.rank)
            tag = (id(self.generator), int(base))
            hit = self._rank_gens.get(key)
            if hit is None or hit[0] != tag:
                gen = torch.Generator(device=device)
                gen.manual_seed(tdist.rank_seed(base, plan.rank))
                self._rank_gens[key] = (tag, gen)
            else:
                gen = hit[1]
        return torch.rand(shape, dtype=torch.float32, device=device, generator=gen)

    # ------------------------------------------------------------------ reference API
    def rollout_mc_search(self, sentences, given_num, uniforms=None):
        """utils/rollout.py:25-55: keep ``given_num`` tokens, sample the rest. Returns [B, L] int64."""
        self._check(sentences)
        B, L = sentences.shape
        if not (1 <= given_num <= L):
            raise RuntimeError("given_num must be in [1, %d]" % L)
        with torch.no_grad():
            u = uniforms if uniforms is not None else self._rand((1, L, L, B), sentences.device)
            comp = ops.gen_rollout(self.gen, sentences, 1, u, g_lo=given_num, g_hi=given_num,
                                   precision=self.gen_precision)
        return comp[0, given_num - 1]

    def get_reward(self, sentences, rollout_num, dis, current_k=0, uniforms=None, dropout_uniforms=None,
                   return_parts=False):
        """utils/rollout.py:128-154.

        uniforms          optional [R,L,L,B] pre-drawn U[0,1) (index [i, given_num-1, t, b]); drawn with
                          torch.rand when omitted.  With a process group (ROLLOUT.group) an explicit tensor is
                          the GLOBAL one, identical on every rank; omitted, every rank draws its own rows from a
                          private generator seeded from (seed, rank).
        dropout_uniforms  optional [R*L, B, F] for the discriminator's live dropout (the reference never
                          calls dis.eval(), so dropout is active whenever dis.training is True)."""
        self._check(sentences)
        R = int(rollout_num)
        B, L = sentences.shape
        dev = sentences.device
        plan = tdist.ShardPlan.from_group(R, self.group)
        Rl = plan.i_hi - plan.i_lo
        with torch.no_grad():
            if plan.world > 1 and self.sync_weights:
                self._sync(dis)
            if uniforms is None:
                u = self._rand((Rl, L, L, B), dev, plan)
            else:
                u = uniforms.reshape(R, L, L, B)[plan.i_lo:plan.i_hi].contiguous()
            raw_local = torch.empty(Rl * L, B, dtype=torch.float32, device=dev)
            comp = torch.empty(0, L, B, L, dtype=torch.int64, device=dev)   # a rank that owns no rollout index
            if Rl > 0:
                p = float(dis.dropout.p) if (dis.training and dis.dropout.p > 0) else 0.0
                weights = _lib.dis_weights(dis, self.dis_precision)
                F = weights[0].F
                du_all = None
                if p > 0 and dropout_uniforms is not None:
                    du_all = dropout_uniforms.reshape(R * L * B, F)[plan.i_lo * L * B:plan.i_hi * L * B]
                # whole rollouts per scorer call, at most dis_chunk sequences each
                step = max(1, self.dis_chunk // max(1, L * B))
                prepared = None
                overlap = self.overlap_prepare and step >= Rl
                if overlap:
                    # one scorer call: its preparation goes to a second stream and runs beside the generator.  That
                    # stream starts behind everything the current one holds so far (the parameters, the sentences,
                    # whoever held the workspace's memory before); the generator is enqueued first, so its persistent
                    # workgroups are on the chip before the preparation's small kernels ask for room.
                    cur = torch.cuda.current_stream(dev)
                    side = self._side.get(dev)
                    if side is None:
                        side = self._side[dev] = torch.cuda.Stream(dev)
                    if du_all is not None:
                        du_all = du_all.contiguous()
                    fork = torch.cuda.Event()
                    fork.record(cur)
                with self._timed("gen_rollout"):
                    comp = ops.gen_rollout(self.gen, sentences, Rl, u, precision=self.gen_precision)
                if overlap:
                    side.wait_event(fork)
                    with torch.cuda.stream(side):
                        prepared = ops.dis_score_rollout_prepare(dis, sentences, Rl, dropout_uniforms=du_all,
                                                                 weights=weights)
                    cur.wait_stream(side)
                    prepared.record_stream(cur)
                with self._timed("dis_score"):
                    for lo in range(0, Rl, step):
                        hi = min(Rl, lo + step)
                        du, seed = None, None
                        if p > 0 and du_all is not None:
                            du = du_all[lo * L * B:hi * L * B].contiguous()
                        elif p > 0 and self.dropout_rng == "torch":
                            du = self._rand(((hi - lo) * L * B, F), dev, plan)
                        elif p > 0:
                            seed = _lib.next_dropout_seed(plan.rank)
                        ops.dis_score_rollout(dis, sentences, comp[lo:hi], col=current_k + 1, dropout_p=p,
                                              dropout_uniforms=du, reward_out=raw_local[lo * L:hi * L],
                                              weights=weights, dropout_seed=seed, prepared=prepared)
            with self._timed("exchange"):
                raw = tdist.gather_rows(raw_local, plan, L, self.group)
            with self._timed("reduce"):
                rewards = ops.reward_reduce(raw, B, L, R)
        if return_parts:
            return rewards, raw, comp
        return rewards


    def get_token_reward(self, sentences, rollout_num, dis, current_k, given_num, uniforms=None):
        """utils/rollout.py:186-204 (no caller in the reference): mean over rollout_num completions from one
        given_num of softmax(dis(samples))[:, current_k + 1].  Returns [B] on the input's device."""
        self._check(sentences)
        R = int(rollout_num)
        B, L = sentences.shape
        if not (1 <= given_num <= L):
            raise RuntimeError("given_num must be in [1, %d]" % L)
        with torch.no_grad():
            u = uniforms if uniforms is not None else self._rand((R, L, L, B), sentences.device)
            comp = ops.gen_rollout(self.gen, sentences, R, u, g_lo=given_num, g_hi=given_num,
                                   precision=self.gen_precision)[:, given_num - 1].contiguous()
            p = float(dis.dropout.p) if (dis.training and dis.dropout.p > 0) else 0.0
            r = ops.dis_score(dis, comp.view(R * B, L), col=current_k + 1, dropout_p=p,
                              precision=self.dis_precision)["reward"].view(R, B)
            return torch.sum(r, dim=0) / R

    def get_reward_csgan(self, target, rollout_num, csgan_clas):
        """utils/rollout.py:206-207: an empty stub in the reference as well."""
        return None

    # ------------------------------------------------------------------ LeakGAN
    def _check_leak(self):
        if getattr(self.gen, "name", None) != "leakgan" or self.step_size < 1:
            raise RuntimeError("textgan_hip ROLLOUT: the LeakGAN rollout needs a LeakGAN_G generator (gen.name == 'leakgan')")

    def rollout_mc_search_leakgan(self, sentences, dis, given_num, uniforms=None):
        """utils/rollout.py:57-126: keep ``given_num`` tokens, complete the rest with LeakGAN_G.forward fed by
        dis.get_feature at every position. Returns [B, L] int64.  uniforms: optional [1,1,L,B]."""
        self._check(sentences)
        self._check_leak()
        B, L = sentences.shape
        if not (1 <= given_num <= L):
            raise RuntimeError("given_num must be in [1, %d]" % L)
        with torch.no_grad():
            u = uniforms if uniforms is not None else self._rand((1, 1, L, B), sentences.device)
            comp = ops.leak_rollout(self.gen, dis, sentences, 1, [given_num], u, dis_precision=self.dis_precision,
                                    precision=self.gen_precision)
        return comp[0, 0]

    def get_reward_leakgan(self, sentences, rollout_num, dis, current_k, uniforms=None, dropout_uniforms=None,
                           return_parts=False):
        """utils/rollout.py:156-184: rewards [B, L // step_size] of the prefixes of length t*step_size + 1.

        uniforms          optional [R, T, L, B] pre-drawn U[0,1) (index [i, t, position, b]), T = L // step_size; the
                          GLOBAL tensor when a process group shards the rollout index.
        dropout_uniforms  optional [R*T, B, F] for the discriminator's live dropout in the final scoring
                          (dis.get_feature inside the rollout has none, models/discriminator.py:51-64)."""
        self._check(sentences)
        self._check_leak()
        R = int(rollout_num)
        B, L = sentences.shape
        T = L // self.step_size
        if T < 1:
            raise RuntimeError("textgan_hip ROLLOUT: max_seq_len %d shorter than step_size %d" % (L, self.step_size))
        given = [t * self.step_size + 1 for t in range(T)]
        dev = sentences.device
        plan = tdist.ShardPlan.from_group(R, self.group)
        Rl = plan.i_hi - plan.i_lo
        with torch.no_grad():
            if plan.world > 1 and self.sync_weights:
                self._sync(dis)
            if uniforms is None:
                u = self._rand((Rl, T, L, B), dev, plan)
            else:
                u = uniforms.reshape(R, T, L, B)[plan.i_lo:plan.i_hi].contiguous()
            raw_local = torch.empty(Rl * T, B, dtype=torch.float32, device=dev)
            comp = torch.empty(0, T, B, L, dtype=torch.int64, device=dev)   # a rank that owns no rollout index
            if Rl > 0:
                weights = _lib.dis_weights(dis, self.dis_precision)
                F = weights[0].F
                p = float(dis.dropout.p) if (dis.training and dis.dropout.p > 0) else 0.0
                du, seed = None, None
                if p > 0 and dropout_uniforms is not None:
                    du = dropout_uniforms.reshape(R * T * B, F)[plan.i_lo * T * B:plan.i_hi * T * B].contiguous()
                elif p > 0 and self.dropout_rng == "torch":
                    du = self._rand((Rl * T * B, F), dev, plan)
                elif p > 0:
                    seed = _lib.next_dropout_seed(plan.rank)
                prepared = None
                if self.overlap_prepare:
                    # as in get_reward: the final scoring's preparation on a second stream, beside the rollout
                    cur = torch.cuda.current_stream(dev)
                    side = self._side.get(dev)
                    if side is None:
                        side = self._side[dev] = torch.cuda.Stream(dev)
                    fork = torch.cuda.Event()
                    fork.record(cur)
                with self._timed("leak_rollout"):
                    comp = ops.leak_rollout(self.gen, dis, sentences, Rl, given, u, dis_weights=weights,
                                            precision=self.gen_precision)
                if self.overlap_prepare:
                    side.wait_event(fork)
                    with torch.cuda.stream(side):
                        prepared = ops.dis_score_prepare(dis, Rl * T * B, L, dev, dropout_uniforms=du, weights=weights)
                    cur.wait_stream(side)
                    prepared.record_stream(cur)
                with self._timed("dis_score"):
                    ops.dis_score(dis, comp.view(Rl * T * B, L), col=current_k + 1, dropout_p=p, dropout_uniforms=du,
                                  reward_out=raw_local.view(-1), weights=weights, dropout_seed=seed, prepared=prepared)
            with self._timed("exchange"):
                raw = tdist.gather_rows(raw_local, plan, T, self.group)
            with self._timed("reduce"):
                rewards = ops.reward_reduce(raw, B, T, R)   # rewards.view(B, T, R).mean(-1) (rollout.py:182-183)
        if return_parts:
            return rewards, raw, comp
        return rewards


class _Timed:
    def __init__(self, timers, name):
        self.timers, self.name = timers, name

    def __enter__(self):
        if self.timers is not None:
            self.s = torch.cuda.Event(enable_timing=True)
            self.e = torch.cuda.Event(enable_timing=True)
            self.s.record()
        return self

    def __exit__(self, *a):
        if self.timers is not None:
            self.e.record()
            self.timers.setdefault(self.name, []).append((self.s, self.e))
        return False


def install(rollout_module, generator_class=None, nll_class=None, pg_classes=()):
    """Replace ``ROLLOUT`` in the reference's ``utils.rollout`` module (and thereby in every
    instructor that does ``from utils.rollout import ROLLOUT`` afterwards).  Optionally bind the same-kernel
    surfaces too, all by attribute on the reference's own classes (no reference file is edited):

      generator_class  models.generator.LSTMGenerator: ``sample(num_samples, batch_size, start_letter)``
                       (generator.py:62) -> tg.generator.sample (CPU int64 [n, L], as the reference returns)
      nll_class        metrics.nll.NLL: the static ``cal_nll(model, data_loader, criterion, gpu)`` (nll.py:50)
                       -> tg.nll.cal_nll
      pg_classes       SeqGAN_G / SentiGAN_G (``batchPGLoss``) and MaliGAN_G (``adv_loss``): the policy-gradient loss
                       with the fused forward / backward (tg.pgloss)

    The replaced attributes are kept under ``_tg_original_<name>`` on the class."""
    rollout_module.ROLLOUT = ROLLOUT
    from . import generator as _generator, nll as _nll, pgloss as _pgloss

    def keep(cls, name):
        if hasattr(cls, name) and not hasattr(cls, "_tg_original_" + name):
            setattr(cls, "_tg_original_" + name, cls.__dict__.get(name, getattr(cls, name)))

    if generator_class is not None:
        keep(generator_class, "sample")

        def sample(self, num_samples, batch_size, start_letter=1):
            return _generator.sample(self, num_samples, batch_size, start_letter)
        generator_class.sample = sample
    if nll_class is not None:
        keep(nll_class, "cal_nll")

        def cal_nll(model, data_loader, criterion=None, gpu=True):
            return _nll.cal_nll(model, data_loader, criterion, gpu)
        nll_class.cal_nll = staticmethod(cal_nll)
    for cls in pg_classes:
        for name, fn in (("batchPGLoss", _pgloss.batchPGLoss), ("adv_loss", _pgloss.adv_loss)):
            if name in cls.__dict__:
                keep(cls, name)
                setattr(cls, name, (lambda f: lambda self, inp, target, reward: f(self, inp, target, reward))(fn))
    return ROLLOUT

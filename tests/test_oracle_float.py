"""Self-checks of the oracle's floating-point restatement (TF1 graphs cannot run here:
parity unpinned): gradients against torch-CPU autograd of the reference's
loss formulas, TF1 sparse-Adam against a hand-computed example with duplicate indices."""
import numpy as np
import pytest

from oracle import graphgan_oracle as orc

torch = pytest.importorskip("torch")


def _dense_grads(n, d, u, v, gu, gv, gb):
    GE = np.zeros((n, d), dtype=np.float64)
    Gb = np.zeros(n, dtype=np.float64)
    np.add.at(GE, u, gu.astype(np.float64))
    np.add.at(GE, v, gv.astype(np.float64))
    np.add.at(Gb, v, gb.astype(np.float64))
    return GE, Gb


def test_discriminator_gradients_match_autograd():
    rs = np.random.RandomState(0)
    n, d, B = 30, 7, 64
    E0 = (rs.randn(n, d) * 0.7).astype(np.float32)
    dis = orc.Discriminator(E0, 1e-3)
    dis.b[:] = rs.randn(n) * 0.2
    u, v = rs.randint(0, n, B), rs.randint(0, n, B)
    y = (rs.rand(B) < 0.5).astype(np.float32)
    lam = 1e-2
    loss, gu, gv, gb = dis.loss_and_grads(u, v, y, lam)
    E = torch.tensor(dis.E, dtype=torch.float64, requires_grad=True)
    b = torch.tensor(dis.b, dtype=torch.float64, requires_grad=True)
    eu, ev, bv = E[torch.tensor(u)], E[torch.tensor(v)], b[torch.tensor(v)]
    s = (eu * ev).sum(1) + bv
    # discriminator.py:26-30: reduce_sum(sigmoid_cross_entropy_with_logits) + lambda*(l2_loss(ev)+l2_loss(eu)+l2_loss(bias))
    L = torch.nn.functional.binary_cross_entropy_with_logits(s, torch.tensor(y, dtype=torch.float64), reduction="sum") \
        + lam * 0.5 * ((ev ** 2).sum() + (eu ** 2).sum() + (bv ** 2).sum())
    L.backward()
    GE, Gb = _dense_grads(n, d, u, v, gu, gv, gb)
    assert abs(loss - L.item()) < 1e-3
    assert np.allclose(GE, E.grad.numpy(), rtol=1e-4, atol=1e-5)
    assert np.allclose(Gb, b.grad.numpy(), rtol=1e-4, atol=1e-5)


def test_generator_gradients_match_autograd():
    rs = np.random.RandomState(1)
    n, d, B = 30, 7, 64
    E0 = (rs.randn(n, d) * 0.7).astype(np.float32)
    gen = orc.Generator(E0, 1e-3)
    gen.b[:] = rs.randn(n) * 0.2
    gen.E[3] *= 40  # drive one score far enough that sigmoid clips at 1e-5 (zero-gradient branch)
    u, v = rs.randint(0, n, B), rs.randint(0, n, B)
    u[0], v[0] = 3, 3
    gen.E[4] = -gen.E[3]
    u[1], v[1] = 3, 4
    r = (rs.rand(B) * 3).astype(np.float32)
    lam = 1e-2
    loss, gu, gv, gb = gen.loss_and_grads(u, v, r, lam)
    E = torch.tensor(gen.E, dtype=torch.float64, requires_grad=True)
    b = torch.tensor(gen.b, dtype=torch.float64, requires_grad=True)
    eu, ev, bv = E[torch.tensor(u)], E[torch.tensor(v)], b[torch.tensor(v)]
    s = (eu * ev).sum(1) + bv
    # generator.py:26-29
    p = torch.clamp(torch.sigmoid(s), 1e-5, 1)
    L = -(torch.log(p) * torch.tensor(r, dtype=torch.float64)).mean() + lam * 0.5 * ((ev ** 2).sum() + (eu ** 2).sum())
    L.backward()
    GE, Gb = _dense_grads(n, d, u, v, gu, gv, gb)
    assert abs(loss - L.item()) < 1e-3 * max(1, abs(L.item()))
    assert np.allclose(GE, E.grad.numpy(), rtol=1e-3, atol=1e-5)
    assert np.allclose(Gb, b.grad.numpy(), rtol=1e-3, atol=1e-5)
    assert gb[1] == 0.0  # clipped pair contributes nothing


def test_reward_formula():
    rs = np.random.RandomState(2)
    E0 = (rs.randn(20, 5) * 2).astype(np.float32)
    dis = orc.Discriminator(E0, 1e-3)
    dis.b[:] = rs.randn(20)
    u, v = rs.randint(0, 20, 100), rs.randint(0, 20, 100)
    s = np.clip((dis.E[u].astype(np.float64) * dis.E[v]).sum(1) + dis.b[v], -10, 10)
    # fp32 log(1 + exp(s)) as the reference computes it: absolute error ~1 ulp of 1.0 near s = -10
    assert np.allclose(dis.reward(u, v), np.log1p(np.exp(s)), rtol=1e-5, atol=1e-6)


def test_tf1_adam_hand_computed_with_duplicates():
    """3 steps on a [4, 1] variable, row 1 duplicated in step 1.  Never-touched rows have
    m = v = 0 and stay put; a row touched ONCE keeps moving on later steps because m and v are
    decayed and applied over all rows -- the TF1.8 sparse-apply behaviour."""
    lr, b1, b2, eps = 0.1, 0.9, 0.999, 1e-8
    var = np.array([[1.0], [2.0], [3.0], [4.0]], dtype=np.float32)
    opt = orc.TF1Adam([var.shape], lr, b1, b2, eps)
    ref = var.astype(np.float64).copy()
    m = np.zeros_like(ref)
    v = np.zeros_like(ref)
    steps = [(np.array([1, 1, 0]), np.array([[0.5], [0.25], [-1.0]])),
             (np.array([2]), np.array([[2.0]])),
             (np.array([0, 2]), np.array([[1.0], [-2.0]]))]
    for t, (idx, g) in enumerate(steps, 1):
        opt.step([var], [(idx, g.astype(np.float32))])
        G = np.zeros_like(ref)
        np.add.at(G, idx, g)
        m = b1 * m + (1 - b1) * G
        v = b2 * v + (1 - b2) * G * G
        lr_t = lr * np.sqrt(1 - b2 ** t) / (1 - b1 ** t)
        ref = ref - lr_t * m / (np.sqrt(v) + eps)
        assert np.allclose(var, ref, rtol=2e-6, atol=1e-7), t
    assert var[3, 0] == 4.0            # never touched: never moves
    # row 1 was touched only in step 1 but moved again in steps 2 and 3 (dense m decay)
    one_step = 2.0 - 0.1 * 1.0  # first step of Adam moves by ~lr
    assert var[1, 0] < one_step - 0.05


def test_lazy_adam_only_moves_touched_rows():
    var = np.ones((4, 2), dtype=np.float32)
    opt = orc.TF1Adam([var.shape], 0.1, lazy=True)
    opt.step([var], [(np.array([1]), np.ones((1, 2), np.float32))])
    after1 = var.copy()
    opt.step([var], [(np.array([2]), np.ones((1, 2), np.float32))])
    assert np.array_equal(var[1], after1[1]) and var[2, 0] < 1.0 and var[0, 0] == 1.0


def test_tf1_adam_dense_equals_torch_adam_with_rescaled_eps():
    """Independent pin of the dense TF1-Adam restatement (a12).  TF 1.8 moves a variable by
    lr * sqrt(1 - b2^t) / (1 - b1^t) * m / (sqrt(v) + eps); torch.optim.Adam by lr / (1 - b1^t) * m / (sqrt(v) / sqrt(1 - b2^t) + eps').
    The two are the same map for eps' = eps / sqrt(1 - b2^t), so a whole training trajectory of the oracle's optimizer
    (sparse slices with duplicates, all rows decaying every step) must equal torch's Adam fed with the densified
    gradients and that per-step eps -- 40 steps of the discriminator's real loss through torch autograd in fp64."""
    rs = np.random.RandomState(3)
    n, d, B, lam = 40, 9, 64, 1e-5
    E0 = (rs.randn(n, d) * 0.5).astype(np.float32)
    dis = orc.Discriminator(E0, 1e-3)
    E = torch.tensor(E0.astype(np.float64), requires_grad=True)
    b = torch.zeros(n, dtype=torch.float64, requires_grad=True)
    opt = torch.optim.Adam([E, b], lr=1e-3, betas=(0.9, 0.999), eps=1e-8)
    for t in range(1, 41):
        u, v = rs.randint(0, n, B), rs.randint(0, n // 2, B)   # rows >= n/2 are touched rarely: decay-only steps matter
        u[:16] = u[0]
        y = (rs.rand(B) < 0.5).astype(np.float32)
        dis.d_step(u, v, y, lam)
        ut, vt, yt = torch.tensor(u), torch.tensor(v), torch.tensor(y.astype(np.float64))
        s = (E[ut] * E[vt]).sum(1) + b[vt]
        loss = torch.nn.functional.binary_cross_entropy_with_logits(s, yt, reduction="sum")
        loss = loss + lam * 0.5 * ((E[vt] ** 2).sum() + (E[ut] ** 2).sum() + (b[vt] ** 2).sum())
        opt.zero_grad()
        loss.backward()
        for g in opt.param_groups:
            g["eps"] = 1e-8 / np.sqrt(1.0 - 0.999 ** t)
        opt.step()
        # fp32 oracle vs fp64 torch: the trajectories stay together to fp32 rounding (a first Adam step is lr * sign(g),
        # so elements whose gradient is ~0 may differ by up to 2 lr in principle; none do with this seed)
        assert np.abs(dis.E - E.detach().numpy()).max() < 2e-5, t
        assert np.abs(dis.b - b.detach().numpy()).max() < 2e-5, t
    assert np.abs(dis.E - E0).max() > 5e-3   # the tables moved


class _TorchModel:
    """The reference's TF graphs written from generator.py / discriminator.py themselves as torch-CPU autograd
    graphs -- losses on the GATHERED rows, gradients by backward(), no hand-derived formula anywhere -- plus a hand-rolled
    tf.train.AdamOptimizer (TF 1.8: m, v and the variable move over ALL rows each step, epsilon outside the bias
    correction, beta powers stepped once per step).  Same interface as the numpy oracle's models (E, b, score,
    d_step / g_step / reward), so that GraphGANOracle can drive it through the reference's schedule."""

    def __init__(self, emb_init, lr, which):
        self.which = which
        self.Et = torch.tensor(np.asarray(emb_init, dtype=np.float32), requires_grad=True)   # embedding_matrix (:11-14)
        self.bt = torch.zeros(self.Et.shape[0], dtype=torch.float32, requires_grad=True)      # bias_vector (:15)
        self.lr, self.b1, self.b2, self.eps = lr, 0.9, 0.999, 1e-8
        self.m = [torch.zeros_like(self.Et), torch.zeros_like(self.bt)]
        self.v = [torch.zeros_like(self.Et), torch.zeros_like(self.bt)]
        self.b1p, self.b2p = torch.tensor(0.9, dtype=torch.float32), torch.tensor(0.999, dtype=torch.float32)

    E = property(lambda self: self.Et.detach().numpy())
    b = property(lambda self: self.bt.detach().numpy())

    def _score(self, u, v):
        eu, ev = self.Et[torch.as_tensor(u)], self.Et[torch.as_tensor(v)]        # tf.nn.embedding_lookup
        return eu, ev, self.bt[torch.as_tensor(v)], None

    def _adam(self, loss):
        self.Et.grad = self.bt.grad = None
        loss.backward()
        with torch.no_grad():
            lr_t = self.lr * torch.sqrt(1 - self.b2p) / (1 - self.b1p)
            for p, m, v in zip((self.Et, self.bt), self.m, self.v):
                g = p.grad if p.grad is not None else torch.zeros_like(p)
                m.mul_(self.b1).add_((1 - self.b1) * g)
                v.mul_(self.b2).add_((1 - self.b2) * g * g)
                p.sub_(lr_t * m / (torch.sqrt(v) + self.eps))
            self.b1p = self.b1p * self.b1
            self.b2p = self.b2p * self.b2

    def reward(self, u, v):  # discriminator.py:33-34
        with torch.no_grad():
            eu, ev, bv, _ = self._score(u, v)
            s = torch.clamp((eu * ev).sum(1) + bv, -10, 10)
            return torch.log(1 + torch.exp(s)).numpy()

    def d_step(self, u, v, label, lam):  # discriminator.py:26-32
        eu, ev, bv, _ = self._score(u, v)
        s = (eu * ev).sum(1) + bv
        ce = torch.nn.functional.binary_cross_entropy_with_logits(s, torch.as_tensor(label, dtype=torch.float32), reduction="sum")
        self._adam(ce + lam * 0.5 * ((ev ** 2).sum() + (eu ** 2).sum() + (bv ** 2).sum()))

    def g_step(self, u, v, reward, lam):  # generator.py:25-31
        eu, ev, bv, _ = self._score(u, v)
        prob = torch.clamp(torch.sigmoid((eu * ev).sum(1) + bv), 1e-5, 1)
        loss = -(torch.log(prob) * torch.as_tensor(reward, dtype=torch.float32)).mean() + lam * 0.5 * ((ev ** 2).sum() + (eu ** 2).sum())
        self._adam(loss)


def test_torch_autograd_trainer_reproduces_the_oracle_epoch():
    """One outer epoch of the reference's schedule on CA-GrQc (shortened to 2 + 2 inner passes: ~15 600 optimizer steps,
    B = 64, dense TF1 Adam) driven twice through the SAME sampler, shuffles and data -- the generator's tables do not move
    before the G-mode walks of the epoch, so both runs see identical walks: once with the numpy oracle's hand-derived gradients
    and sparse-apply Adam, once with the torch autograd restatement above.  If the oracle's float half mis-stated the
    reference's graphs (a wrong gradient term, a wrong Adam detail), the two would part; they agree to rounding, and BOTH move
    the generator's link-prediction accuracy the same way from the shipped embeddings' 0.7598 -- what the engine reproduces
     is what these losses do, not an artefact of the restatement."""
    from tests.helpers import load_ca_grqc, ca_grqc_init_embeddings
    d, n, graph = load_ca_grqc()
    emb = ca_grqc_init_embeddings(d, n)
    cfg = orc.Config()
    cfg.n_epochs_dis, cfg.n_epochs_gen, cfg.dis_interval, cfg.gen_interval = 2, 2, 2, 2
    runs = []
    for flavour in ("numpy", "torch"):
        o = orc.GraphGANOracle(n, graph, emb, emb, cfg=cfg, rng="counter", arith="spec", seed=7)
        if flavour == "torch":
            o.generator = _TorchModel(emb, cfg.lr_gen, 0)
            o.discriminator = _TorchModel(emb, cfg.lr_dis, 1)
        o.train_epoch(0)
        acc = [orc.eval_link_prediction(np.asarray(m.E, dtype=np.float64), d["test"], d["test_neg"]) for m in (o.generator, o.discriminator)]
        runs.append((np.array(o.generator.E), np.array(o.generator.b), np.array(o.discriminator.E), np.array(o.discriminator.b), acc, dict(o.counters)))
    a, t = runs
    assert a[5] == t[5] and a[5]["hops"] > 290000            # identical walks went in
    for x, y, name in zip(a[:4], t[:4], ("gen E", "gen b", "dis E", "dis b")):
        diff = np.abs(x - y)
        assert diff.mean() < 2e-5 and np.quantile(diff, 0.999) < 1e-3, (name, diff.mean(), diff.max())
    acc0 = orc.eval_link_prediction(emb, d["test"], d["test_neg"])
    assert abs(acc0 - 0.7598343685300207) < 1e-12
    print("accuracy gen / dis: start %.4f, numpy oracle %.4f / %.4f, torch autograd %.4f / %.4f" % (acc0, a[4][0], a[4][1], t[4][0], t[4][1]))
    for k in range(2):
        assert abs(a[4][k] - t[4][k]) <= 0.005                # +-0.5 % absolute
    # ... and the comparison is INFORMATIVE: after this short schedule the generator is far from where it started and far from
    # chance (it gains ~0.1: the long default schedule is what later drives it down), on both sides alike
    assert abs(a[4][0] - acc0) > 0.05 and abs(t[4][0] - acc0) > 0.05 and min(a[4][0], t[4][0]) > 0.6

"""HMC and NUTS for *user* gradient callbacks (host side of the split path).

With Python ``logl_grad`` / ``logp_grad`` callbacks the gradient jumps cannot run inside a kernel; they run here, between
``ptmi_propose`` and ``ptmi_accept`` (``sampler.PTSampler._split_step``).  The behaviour is that of the reference's
``PTMCMCSampler/nutsjump.py`` (cited NJ:<lines>): whitening by the Cholesky factor of the initial covariance (NJ:51-54,
71-90), leapfrog (NJ:149-169), ``HMCJump`` (NJ:238-291), ``NUTSJump`` with dual averaging (NJ:379-463, 654-840).  All
randomness is drawn from the global ``np.random`` state in the reference's order, so a seeded run reproduces the reference's
jumps call by call (tests/test_gradjump.py against tests/golden/gradjump.npz).

The organisation is this package's own and mirrors the device kernel ``csrc/ptmi_gj.inc.h``: phase-space points are
``_Point`` records, and the doubling tree of NUTS (the reference's recursive ``build_tree``, NJ:495-652) is built
*iteratively* -- leaves are generated outwards from the growing end and finished subtrees wait on an explicit stack until
their right sibling of equal height completes, exactly the binary-counter schedule of the kernel.  MALA (NJ:182-235) is not
provided: the reference itself flags it as not working (PTMCMCSampler.py:230-231); the ``Trajectory`` dump (NJ:294-377) is
a debug feature and out of scope.
"""
import numpy as np
import scipy.linalg as sl


class _Point(object):
    """A phase-space point in whitened coordinates: position, momentum, gradient and log-density at the position."""
    __slots__ = ("q", "p", "g", "lp")

    def __init__(self, q, p, g, lp):
        self.q, self.p, self.g, self.lp = q, p, g, lp

    def energy(self):
        """log joint density  lp - p.p/2  (NJ:133-147; a momentum of the wrong shape gives NaN there)."""
        try:
            return self.lp - 0.5 * np.dot(self.p, self.p)
        except ValueError:
            return np.nan


class _Subtree(object):
    """A finished run of 2**height consecutive leapfrog points: its first (innermost) and last (outermost) point,
    the point it proposes, how many of its points lie in the slice, whether it may keep growing, and the acceptance
    statistics that feed the step-size adaptation."""
    __slots__ = ("height", "first", "last", "pick", "inside", "alive", "acc_sum", "acc_cnt")


class GradientJump(object):
    """Whitened target and integrator shared by the gradient jumps."""

    def __init__(self, loglik_grad, logprior_grad, mm_inv, nburn=100):
        self._ll, self._lp = loglik_grad, logprior_grad
        self.mm_inv, self.nburn = mm_inv, nburn
        self.ndim = len(mm_inv)
        # NJ:53-54: L = chol(cov);  x = L^T q,  q = L^-T x,  d/dq = L d/dx
        self.cov_cf = sl.cholesky(mm_inv, lower=True)
        self.cov_cfi = sl.solve_triangular(self.cov_cf, np.eye(self.ndim), trans=0, lower=True)
        self.name = "GradientJUMP"
        self.epsilon, self.beta, self.iter = None, 1.0, 0.0
        print("WARNING: GradientJumps not yet adaptive. Choose cov wisely!")               # NJ:45

    @property
    def __name__(self):
        return self.name

    # -- coordinates
    def forward(self, x):
        return np.dot(self.cov_cfi.T, x)

    def backward(self, q):
        return np.dot(self.cov_cf.T, q)

    def target(self, q):
        """Tempered log-posterior and its gradient in whitened coordinates (NJ:71-90)."""
        x = self.backward(q)
        ll, dll = self._ll(x)
        lp, dlp = self._lp(x)
        return self.beta * ll + lp, np.dot(self.cov_cf, self.beta * dll + dlp)

    def _enter(self, x, beta):
        """Common prologue of a jump call: call counter, argument check, tempering, whitened start point."""
        self.iter += 1
        x = np.atleast_1d(x)
        if x.ndim > 1:
            raise ValueError("x is expected to be a 1-D array")
        self.beta = beta
        q = self.forward(x)
        lp, g = self.target(q)
        return q, g, lp

    def momenta(self):
        return np.random.randn(self.ndim)                                               # NJ:92-94

    def step(self, pt, h):
        """One leapfrog of signed size ``h`` from ``pt`` (NJ:149-169)."""
        half = pt.p + 0.5 * h * pt.g
        q = pt.q + h * half
        lp, g = self.target(q)
        return _Point(q, half + 0.5 * h * g, g, lp)


class HMCJump(GradientJump):
    """A Hamiltonian trajectory of fixed step size and a random number of steps (NJ:238-291)."""

    def __init__(self, loglik_grad, logprior_grad, mm_inv, nburn=100, stepsize=0.1, nminsteps=10, nmaxsteps=300):
        super(HMCJump, self).__init__(loglik_grad, logprior_grad, mm_inv, nburn=nburn)
        self.name = "HMCJump"
        self.epsilon, self.nminsteps, self.nmaxsteps = stepsize, nminsteps, nmaxsteps

    def __call__(self, x, iter, beta):
        q, g, lp = self._enter(x, beta)
        pt = _Point(q, self.momenta(), g, lp)
        e0 = e1 = pt.energy()
        left = np.random.randint(self.nminsteps, self.nmaxsteps)
        while left > 0:
            pt = self.step(pt, self.epsilon)
            e1 = pt.energy()
            left -= 1
            if (e1 - 1000.0) < e0:        # the reference's guard as written (NJ:284-286): ends the walk unless the energy soared
                break
        return self.backward(pt.q), e1 - e0


class _StepSizeAdapter(object):
    """Nesterov dual averaging of log(epsilon) towards a target acceptance (Hoffman & Gelman alg. 6; NJ:410-417, 805-816)."""

    gamma, t0, kappa = 0.05, 10, 0.75

    def __init__(self, delta):
        self.delta, self.mu, self.Hbar, self.log_avg = delta, None, 0, 0.0

    def anchor(self, eps0):
        self.mu = np.log(10.0 * eps0)

    def tell(self, calls, mean_accept):
        w = 1.0 / float(calls + self.t0)
        self.Hbar = (1.0 - w) * self.Hbar + w * (self.delta - mean_accept)

    def propose(self, calls, eps_avg):
        """(next epsilon, new running average) while adapting."""
        eps = np.exp(self.mu - np.sqrt(calls) / self.gamma * self.Hbar)
        w = calls ** -self.kappa
        return eps, np.exp((1.0 - w) * np.log(eps_avg) + w * np.log(eps))


class NUTSJump(GradientJump):
    """No-U-turn jump; the step size is adapted by dual averaging while ``iter <= nburn``."""

    def __init__(self, loglik_grad, logprior_grad, mm_inv, nburn=100, trajectoryDir=None, write_burnin=False,
                 force_trajlen=None, force_epsilon=None, delta=0.6):
        super(NUTSJump, self).__init__(loglik_grad, logprior_grad, mm_inv, nburn=nburn)
        if trajectoryDir is not None:
            raise NotImplementedError("trajectory dumps (nutsjump.py:294-377) are a debug feature and not provided")
        self.name = "NUTSJUMP"
        self.delta = delta
        self._adapt = _StepSizeAdapter(delta)
        self.epsilonbar = 1.0 if force_epsilon is None else force_epsilon
        self.force_trajlen, self.force_epsilon = force_trajlen, force_epsilon

    # the reference's attribute names, read by callers and tests
    @property
    def mu(self):
        return self._adapt.mu

    @property
    def Hbar(self):
        return self._adapt.Hbar

    # -- pieces
    def _first_epsilon(self, start):
        """Heuristic initial step (NJ:435-463): shrink until the first leapfrog is finite, then move by factors of two
        until the acceptance of a single step crosses 1/2."""
        origin = _Point(start.q, self.momenta(), start.g, start.lp)
        e_origin = origin.energy()
        eps, shrink = 1.0, 1.0
        trial = self.step(origin, eps)
        while np.isinf(trial.lp) or np.isinf(trial.g).any():
            shrink *= 0.5
            trial = self.step(origin, eps * shrink)
        eps = 0.5 * shrink * eps
        ratio = np.exp(trial.energy() - e_origin)
        sgn = 2.0 * float(ratio > 0.5) - 1.0
        while (ratio ** sgn) > (2.0 ** (-sgn)):
            eps = eps * (2.0 ** sgn)
            ratio = np.exp(self.step(origin, eps).energy() - e_origin)
        return eps

    def _turning(self, minus, plus, walked):
        """True while the trajectory between its two ends has not turned back (NJ:465-493); with a forced length it is
        the leaf count that decides."""
        if self.force_trajlen is not None:
            return walked < self.force_trajlen
        span = plus.q - minus.q
        return (np.dot(span, minus.p) >= 0) & (np.dot(span, plus.p) >= 0)

    @staticmethod
    def _leaf(pt, logu, e0):
        e = pt.energy()
        t = _Subtree()
        t.height, t.first, t.last, t.pick = 0, pt, pt, pt
        t.inside = int(logu < e)                     # in the slice
        t.alive = int((logu - 1000.0) < e)           # not hopelessly off the energy shell
        t.acc_sum, t.acc_cnt = min(1.0, np.exp(e - e0)), 1
        return t

    def _join(self, a, b, sign, walked):
        """Subtree ``b`` was grown right after ``a`` in direction ``sign``: their union (NJ:545-650)."""
        if np.random.uniform() < (float(b.inside) / max(float(int(a.inside) + int(b.inside)), 1.0)):
            a.pick = b.pick
        a.last = b.last
        a.inside = int(a.inside) + int(b.inside)
        ends = (a.first, a.last) if sign == 1 else (a.last, a.first)
        a.alive = int(a.alive and b.alive and self._turning(ends[0], ends[1], walked))
        a.acc_sum, a.acc_cnt = a.acc_sum + b.acc_sum, a.acc_cnt + b.acc_cnt
        a.height = max(a.height, b.height) + 1
        return a

    def _grow(self, edge, sign, height, eps, logu, e0, walked):
        """2**height further leapfrogs beyond ``edge`` as one subtree.  Finished subtrees wait on ``pending`` (heights
        strictly decreasing towards the top) for a right sibling of their height; a subtree that may not grow any
        further ends the build, and what is pending absorbs it from the top down -- which is what the recursion does
        when it returns a stopped subtree upwards."""
        pending, cur, made, tree = [], edge, 0, None
        while made < (1 << height):
            cur = self.step(cur, sign * eps)
            made += 1
            tree = self._leaf(cur, logu, e0)
            while pending and pending[-1].height == tree.height and pending[-1].alive:
                tree = self._join(pending.pop(), tree, sign, walked + made)
            if not tree.alive:
                break
            pending.append(tree)
            tree = None
        if tree is None:
            tree = pending.pop()
        while pending:
            tree = self._join(pending.pop(), tree, sign, walked + made)
        return tree, walked + made

    def __call__(self, x, iter, beta):
        q, g, lp = self._enter(x, beta)
        start = _Point(q, None, g, lp)
        if self.epsilon is None:
            self.epsilon = self._first_epsilon(start) if self.force_epsilon is None else self.force_epsilon
            self._adapt.anchor(self.epsilon)
        elif self.force_epsilon is not None:
            self.epsilon = self.force_epsilon
        start.p = self.momenta()
        e0 = start.energy()
        logu = float(e0 - np.random.exponential(1, size=1)[0])                           # slice variable, NJ:718
        minus = plus = start
        chosen, total, height, alive = start, 1, 0, 1
        walked = {1: 0, -1: 0}               # the reference's ip / im leaf indices (they share one odometer, NJ:617-650)
        while alive == 1:
            sign = int(2 * (np.random.uniform() < 0.5) - 1)
            sub, end = self._grow(plus if sign == 1 else minus, sign, height, self.epsilon, logu, e0, walked[sign])
            walked[sign], walked[-sign] = end, end - 1
            if sign == 1:
                plus = sub.last
            else:
                minus = sub.last
            if (sub.alive == 1) and (np.random.uniform() < min(1, float(sub.inside) / float(total))):
                chosen = sub.pick
            total += sub.inside
            alive = sub.alive and self._turning(minus, plus, max(walked[1], walked[-1]))
            height += 1
        if self.force_epsilon is None:
            self._adapt.tell(self.iter, sub.acc_sum / float(sub.acc_cnt))
            if iter <= self.nburn:
                self.epsilon, self.epsilonbar = self._adapt.propose(self.iter, self.epsilonbar)
            else:
                self.epsilon = self.epsilonbar
        # NUTS is its own accept step: qxy cancels the outer Hastings ratio (NJ:838)
        return self.backward(chosen.q), lp - chosen.lp

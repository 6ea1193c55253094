"""Gradient-based jumps for the host-callback path: HMC and NUTS.

Behavioural restatement of the reference's ``PTMCMCSampler/nutsjump.py`` (cited NJ:<lines>):
``GradientJump`` whitening (NJ:51-54, 71-90), ``leapfrog`` (NJ:149-169), ``HMCJump`` (NJ:238-291)
and ``NUTSJump`` -- Hoffman & Gelman (2011) algorithm 6 with dual averaging of the step size
(NJ:379-463, 465-493, 495-652, 654-840).  They need the user's Python gradient callbacks, so
they run on the host between the propose and accept kernels (``sampler.PTSampler._split_step``).
All randomness comes from the global ``np.random`` state in the reference's draw order, so a
seeded run reproduces the reference's jumps (tests/test_gradjump.py against
tests/golden/gradjump.npz).  ``MALAJump`` (NJ:182-235) is included for signature completeness although the
reference flags it as not working properly (PTMCMCSampler.py:230-231).  The ``Trajectory`` debug buffer
(NJ:294-377) is out of scope.
"""
import numpy as np
import scipy.linalg as sl


class GradientJump(object):
    """Whitened log-probability and leapfrog integrator shared by the gradient jumps."""

    def __init__(self, loglik_grad, logprior_grad, mm_inv, nburn=100):
        self._loglik_grad, self._logprior_grad = loglik_grad, logprior_grad
        self.mm_inv, self.nburn = mm_inv, nburn
        self.ndim = len(mm_inv)
        self.cov_cf = sl.cholesky(mm_inv, lower=True)                                    # NJ:53
        self.cov_cfi = sl.solve_triangular(self.cov_cf, np.eye(self.ndim), trans=0, lower=True)
        self.name = "GradientJUMP"
        self.epsilon, self.beta, self.iter = None, 1.0, 0.0
        print("WARNING: GradientJumps not yet adaptive. Choose cov wisely!")             # NJ:45

    @property
    def __name__(self):
        return self.name

    def forward(self, x):
        return np.dot(self.cov_cfi.T, x)

    def backward(self, q):
        return np.dot(self.cov_cf.T, q)

    def func_grad_white(self, q):
        """beta*logl + logp and its gradient with respect to the whitened coordinates (NJ:71-90)."""
        x = self.backward(q)
        ll, ll_grad = self._loglik_grad(x)
        lp, lp_grad = self._logprior_grad(x)
        return self.beta * ll + lp, np.dot(self.cov_cf, self.beta * ll_grad + lp_grad)

    def draw_momenta(self):
        return np.random.randn(self.ndim)

    @staticmethod
    def loghamiltonian(logl, r):
        try:
            return logl - 0.5 * np.dot(r, r)
        except ValueError:
            return np.nan

    def leapfrog(self, theta, r, grad, epsilon):
        """One leapfrog step (NJ:149-169): half kick, drift, gradient, half kick."""
        rhalf = r + 0.5 * epsilon * grad
        thetaprime = theta + epsilon * rhalf
        logpprime, gradprime = self.func_grad_white(thetaprime)
        return thetaprime, rhalf + 0.5 * epsilon * gradprime, gradprime, logpprime


class MALAJump(GradientJump):
    """Metropolis-adjusted Langevin step along one whitened coordinate (NJ:182-235)."""

    def __init__(self, loglik_grad, logprior_grad, mm_inv, nburn=100):
        super(MALAJump, self).__init__(loglik_grad, logprior_grad, mm_inv, nburn=nburn)
        self.name = "MALAJump"
        self.cd = 2.4 / np.sqrt(self.ndim)
        self._u, self._s = np.eye(self.ndim), np.ones(self.ndim)       # whitened space: identity decomposition

    def __call__(self, x, iter, beta):
        self.iter += 1
        x = np.atleast_1d(x)
        if x.ndim > 1:
            raise ValueError("x is expected to be a 1-D array")
        self.beta = beta
        q0 = self.forward(x)
        _, grad0 = self.func_grad_white(q0)
        i = np.random.randint(0, self.ndim)
        vec, val = self._u[i, :], self._s[i]
        dist = np.random.randn()
        mq0 = q0 + 0.5 * vec * self.cd ** 2 * np.dot(vec, grad0) / 2 / val
        q1 = mq0 + dist * vec * self.cd / np.sqrt(val)
        _, grad1 = self.func_grad_white(q1)
        mq1 = q1 + 0.5 * vec * self.cd ** 2 * np.dot(vec, grad1) / 2 / val
        qxy = 0.5 * (np.sum((mq0 - q1) ** 2 / val) - np.sum((mq1 - q0) ** 2 / val))
        return self.backward(q1), qxy


class HMCJump(GradientJump):
    """Fixed-step Hamiltonian trajectory of a random number of leapfrogs (NJ:238-291)."""

    def __init__(self, loglik_grad, logprior_grad, mm_inv, nburn=100, stepsize=0.1, nminsteps=10, nmaxsteps=300):
        super(HMCJump, self).__init__(loglik_grad, logprior_grad, mm_inv, nburn=nburn)
        self.name = "HMCJump"
        self.epsilon, self.nminsteps, self.nmaxsteps = stepsize, nminsteps, nmaxsteps

    def __call__(self, x, iter, beta):
        self.iter += 1
        x = np.atleast_1d(x)
        if x.ndim > 1:
            raise ValueError("x is expected to be a 1-D array")
        self.beta = beta
        q = self.forward(x)
        logp0, grad = self.func_grad_white(q)
        p = self.draw_momenta()
        joint0 = self.loghamiltonian(logp0, p)
        nsteps = np.random.randint(self.nminsteps, self.nmaxsteps)
        for _ in range(nsteps):
            q, p, grad, logp1 = self.leapfrog(q, p, grad, self.epsilon)
            joint1 = self.loghamiltonian(logp1, p)
            if (joint1 - 1000.0) < joint0:                # hopelessly inaccurate: stop (NJ:284-286)
                break
        return self.backward(q), joint1 - joint0


class _Tree(object):
    """What one (sub)tree hands to its parent: both ends, the proposed point, counts, flags."""
    __slots__ = ("tm", "rm", "gm", "tp", "rp", "gp", "theta", "grad", "logp", "n", "s", "alpha", "nalpha",
                 "ip", "im")


class NUTSJump(GradientJump):
    """No-U-Turn sampler jump with dual-averaging step-size adaptation during burn-in."""

    def __init__(self, loglik_grad, logprior_grad, mm_inv, nburn=100, trajectoryDir=None, write_burnin=False,
                 force_trajlen=None, force_epsilon=None, delta=0.6):
        super(NUTSJump, self).__init__(loglik_grad, logprior_grad, mm_inv, nburn=nburn)
        if trajectoryDir is not None:
            raise NotImplementedError("trajectory dumps (nutsjump.py:294-377) are a debug feature and not provided")
        self.name = "NUTSJUMP"
        self.delta = delta
        self.gamma, self.t0, self.kappa = 0.05, 10, 0.75                                   # NJ:415-417
        self.mu, self.epsilonbar, self.Hbar = None, 1.0, 0
        self.force_trajlen, self.force_epsilon = force_trajlen, force_epsilon
        if force_epsilon is not None:
            self.epsilonbar = force_epsilon

    def _accept_ratio(self, logpprime, rprime, logp0, r0):
        return np.exp(self.loghamiltonian(logpprime, rprime) - self.loghamiltonian(logp0, r0))

    def find_reasonable_epsilon(self, theta0, grad0, logp0):
        """Heuristic first step size (NJ:435-463): halve until finite, then double/halve across 1/2."""
        epsilon = 1.0
        r0 = self.draw_momenta()
        _, rprime, gradprime, logpprime = self.leapfrog(theta0, r0, grad0, epsilon)
        k = 1.0
        while np.isinf(logpprime) or np.isinf(gradprime).any():
            k *= 0.5
            _, rprime, _, logpprime = self.leapfrog(theta0, r0, grad0, epsilon * k)
        epsilon = 0.5 * k * epsilon
        acceptprob = self._accept_ratio(logpprime, rprime, logp0, r0)
        a = 2.0 * float(acceptprob > 0.5) - 1.0
        while (acceptprob ** a) > (2.0 ** (-a)):
            epsilon = epsilon * (2.0 ** a)
            _, rprime, _, logpprime = self.leapfrog(theta0, r0, grad0, epsilon)
            acceptprob = self._accept_ratio(logpprime, rprime, logp0, r0)
        return epsilon

    def stop_criterion(self, thetaminus, thetaplus, rminus, rplus, force_trajlen, index):
        """True while the trajectory has not made a U-turn (NJ:465-493)."""
        if force_trajlen is not None:
            return index < force_trajlen
        dtheta = thetaplus - thetaminus
        return (np.dot(dtheta, rminus) >= 0) & (np.dot(dtheta, rplus) >= 0)

    def build_tree(self, theta, r, grad, logu, v, j, epsilon, joint0, ind):
        """Height-j subtree in direction v (NJ:495-652).  Returns a _Tree."""
        if j == 0:
            t = _Tree()
            thetaprime, rprime, gradprime, logpprime = self.leapfrog(theta, r, grad, v * epsilon)
            joint = self.loghamiltonian(logpprime, rprime)
            t.n = int(logu < joint)                        # inside the slice
            t.s = int((logu - 1000.0) < joint)             # not wildly inaccurate
            t.tm = t.tp = t.theta = thetaprime
            t.rm = t.rp = rprime
            t.gm = t.gp = t.grad = gradprime
            t.logp = logpprime
            t.alpha, t.nalpha = min(1.0, np.exp(joint - joint0)), 1
            t.ip, t.im = (ind + 1, ind) if v == 1 else (ind, ind + 1)
            return t
        t = self.build_tree(theta, r, grad, logu, v, j - 1, epsilon, joint0, ind)
        if t.s == 1:
            if v == -1:
                u = self.build_tree(t.tm, t.rm, t.gm, logu, v, j - 1, epsilon, joint0, t.im)
                t.tm, t.rm, t.gm = u.tm, u.rm, u.gm
            else:
                u = self.build_tree(t.tp, t.rp, t.gp, logu, v, j - 1, epsilon, joint0, t.ip)
                t.tp, t.rp, t.gp = u.tp, u.rp, u.gp
            t.ip, t.im = u.ip, u.im
            if np.random.uniform() < (float(u.n) / max(float(int(t.n) + int(u.n)), 1.0)):
                t.theta, t.grad, t.logp = u.theta, u.grad, u.logp
            t.n = int(t.n) + int(u.n)
            t.s = int(t.s and u.s and self.stop_criterion(t.tm, t.tp, t.rm, t.rp, self.force_trajlen, max(t.ip, t.im)))
            t.alpha, t.nalpha = t.alpha + u.alpha, t.nalpha + u.nalpha
        return t

    def __call__(self, x, iter, beta):
        self.iter += 1
        x = np.atleast_1d(x)
        if x.ndim > 1:
            raise ValueError("x is expected to be a 1-D array")
        q = self.forward(x)
        self.beta = beta
        logp, grad = self.func_grad_white(q)
        if self.epsilon is None:
            self.epsilon = self.find_reasonable_epsilon(q, grad, logp) if self.force_epsilon is None else self.force_epsilon
            self.mu = np.log(10.0 * self.epsilon)
        elif self.force_epsilon is not None:
            self.epsilon = self.force_epsilon
        r0 = self.draw_momenta()
        joint = self.loghamiltonian(logp, r0)
        logu = float(joint - np.random.exponential(1, size=1)[0])
        sample, lnprob = np.copy(q), np.copy(logp)
        tm = tp = np.copy(q)
        rm = rp = np.copy(r0)
        gm = gp = np.copy(grad)
        j, n, s = 0, 1, 1
        ip = im = 0
        while s == 1:
            v = int(2 * (np.random.uniform() < 0.5) - 1)
            if v == -1:
                t = self.build_tree(tm, rm, gm, logu, v, j, self.epsilon, joint, im)
                tm, rm, gm = t.tm, t.rm, t.gm
            else:
                t = self.build_tree(tp, rp, gp, logu, v, j, self.epsilon, joint, ip)
                tp, rp, gp = t.tp, t.rp, t.gp
            ip, im = t.ip, t.im
            if (t.s == 1) and (np.random.uniform() < min(1, float(t.n) / float(n))):
                sample, lnprob = np.copy(t.theta), np.copy(t.logp)
            n += t.n
            s = t.s and self.stop_criterion(tm, tp, rm, rp, self.force_trajlen, max(ip, im))
            j += 1
        if self.force_epsilon is None:                                                      # NJ:805-816
            eta = 1.0 / float(self.iter + self.t0)
            self.Hbar = (1.0 - eta) * self.Hbar + eta * (self.delta - t.alpha / float(t.nalpha))
            if iter <= self.nburn:
                self.epsilon = np.exp(self.mu - np.sqrt(self.iter) / self.gamma * self.Hbar)
                eta = self.iter ** -self.kappa
                self.epsilonbar = np.exp((1.0 - eta) * np.log(self.epsilonbar) + eta * np.log(self.epsilon))
            else:
                self.epsilon = self.epsilonbar
        # the outer Hastings test must always pass: qxy undoes its ratio (NJ:838)
        return self.backward(sample), logp - lnprob

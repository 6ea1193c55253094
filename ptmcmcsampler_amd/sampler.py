"""``PTSampler``: the reference's class surface on top of the MI355X engine.

Same constructor, ``sample()`` keywords and defaults, ``addProposalToCycle()`` /
``addAuxilaryJump()``, callback signatures, public attributes and output files as
``PTMCMCSampler.PTSampler`` (PTMCMCSampler/PTMCMCSampler.py:40-1067), so a script written for
the reference runs unchanged as ONE process driving one GPU.  Differences, all additive:

* the reference runs one temperature per MPI rank; here one process owns every temperature
  (``ntemps=``) of ``nwalkers=`` independent replicas, and user code sees ``nchain == 1``
  semantics for its callbacks (``comm`` may be the size-1 dummy or ``None``);
* ``logl`` may be a Python callable (evaluated on the host between the propose and accept
  kernels, one launch pair per iteration) **or** a tuple naming a device likelihood --
  ``("iso",)``, ``("dense", mu, P)``, ``("curved",)``, ``("interval", a, b)`` -- and ``logp`` a callable or
  ``("flat",)`` / ``("box", lo, hi)``; with device likelihoods and no host-side jumps the
  fused K-step kernel runs;
* ``logl_grad`` / ``logp_grad`` are the reference's gradient callbacks (HMC / NUTS then run on the host,
  ``gradjump.py``) or, with a device likelihood, ``True`` for its built-in analytic gradient (NUTS / HMC then run
  inside the kernel, PTMCMCSampler.py:225-258 with the same weights and step-size keywords);
* ``batched=True``: ``logl`` / ``logp`` are called once per iteration with the device tensor of all proposals,
  ``f(X[n, ndim]) -> [n]`` (torch in, torch out; loglargs / loglkwargs still apply) -- the same boundary as the reference's
  ``_function_wrapper`` (PTMCMCSampler.py:1072-1086), one call per batch instead of one per chain, nothing copied to the
  host; custom Python jumps cannot be mixed in;
* engine options: ``cov_mode="pooled"`` (one covariance adapted from all walkers instead of one per walker),
  ``swap_mode="oddeven"`` (disjoint swap pairs instead of the reference's hot -> cold sweep), ``pick_mode="walker"`` (one
  proposal-type draw per walker and iteration), ``eig_mode="ql"`` / ``"jacobi"`` / ``"sytrd"`` / ``"hipsolver"`` (covariance epochs factorized on the device: per-walker matrices by
  QL or Jacobi kernels, one large pooled matrix by a one-kernel tridiagonalization + the library's divide-and-conquer, or the ROCm
  library's eigensolver; see PTEngine),
  ``keep_walkers``.

Attributes ``_chain, _lnlike, _lnprob, naccepted, nswap_accepted, swapProposed, jumpDict, cov,
U, S, ladder, temp`` describe walker 0's T = 1 chain, as rank 0's do in the reference.
"""
import os
import sys
import time

import numpy as np

from . import _lib
from .ladder import temperature_ladder


class _DummyComm(object):
    """Size-1 communicator with the duck type of PTMCMCSampler/nompi4py.py:1-37."""

    def Get_rank(self):
        return 0

    def Get_size(self):
        return 1

    def barrier(self):
        pass

    def send(self, obj, dest=1, tag=55):
        pass

    def recv(self, source=1, tag=55):
        pass

    def scatter(self, sendobj, **kwargs):
        return sendobj[0] if sendobj is not None else None

    def bcast(self, obj, **kwargs):
        return obj

    def gather(self, sendobj, **kwargs):
        return [sendobj]


class _function_wrapper(object):
    """Binds args/kwargs so that logl(x) is unary (PTMCMCSampler.py:1072-1086)."""

    def __init__(self, f, args, kwargs):
        self.f, self.args, self.kwargs = f, args, kwargs

    def __call__(self, x):
        return self.f(x, *self.args, **self.kwargs)


class _PerRankJump(object):
    """A stateful gradient jump, one instance per (walker, temperature rank): in the reference every MPI rank
    builds its own HMC / NUTS object (PTMCMCSampler.py:226-258), with its own step size and adaptation state."""

    def __init__(self, sampler, factory):
        import contextlib
        import io
        self.sampler, self.factory, self.inst = sampler, factory, {}
        self.proto = factory()                               # prints the reference's construction warning once
        self.name = self.proto.__name__
        self._quiet = lambda: contextlib.redirect_stdout(io.StringIO())

    @property
    def __name__(self):
        return self.name

    def __call__(self, x, iter, beta):
        key = self.sampler._ctx
        j = self.inst.get(key)
        if j is None:
            if not self.inst:
                j = self.proto
            else:
                with self._quiet():
                    j = self.factory()
            self.inst[key] = j
        return j(x, iter, beta)


# ptmi_checkpoint.npz: 2 = DE rows in the piece-cyclic device format + run fingerprint; 3 = pooled mode keeps ONE (mu, M2);
# 4 = AM rows in the device's row format; 5 = the fingerprint also covers ladder, Tskip, weights, engine modes and likelihood
_CKPT_FORMAT = 5
_CKPT_FORMAT_WHY = {2: "the DE history's row layout", 3: "the pooled statistics' state", 4: "the AM buffer's row layout",
                    5: "what the run fingerprint covers"}


class PTSampler(object):
    def __init__(self, ndim, logl, logp, cov, groups=None, loglargs=[], loglkwargs={}, logpargs=[], logpkwargs={},
                 logl_grad=None, logp_grad=None, comm=None, outDir="./chains", verbose=True, resume=False, seed=None,
                 nwalkers=1, ntemps=None, device=0, cov_mode="per_walker", keep_walkers=1, swap_mode="sweep",
                 pick_mode="chain", eig_mode="lapack", checkpoint=None, batched=False, nuts_maxdepth=24):
        self.comm = comm if comm is not None else _DummyComm()
        if self.comm.Get_size() != 1:
            raise NotImplementedError(
                "one process drives all temperatures here: run without mpirun and pass ntemps=%d" % self.comm.Get_size())
        self.MPIrank, self.nchain = 0, int(ntemps) if ntemps else 1
        self.nwalkers, self.device_index, self.cov_mode = int(nwalkers), device, cov_mode
        self.swap_mode = swap_mode                          # "sweep" = PTswap as the reference; "oddeven" see PTEngine
        self.pick_mode, self.eig_mode = pick_mode, eig_mode # engine options, see PTEngine
        self.nuts_maxdepth = int(nuts_maxdepth)             # device NUTS: tree-height cap (24 = none in practice, as the reference)
        # device checkpoints (ptmi_checkpoint.npz beside the chain file) are written at every save when the run may be
        # resumed: checkpoint=True, or -- by default -- when it was itself started with resume=True.  A run of one chain
        # (ntemps = nwalkers = 1) can also be resumed from its chain file alone, as in the reference (:290-319); a ladder or a
        # batch needs checkpoint=True from its first run on (the device state of a large batch is GBs: opt-in)
        self.checkpoint = bool(resume) if checkpoint is None else bool(checkpoint)
        # batched=True: logl / logp take ALL proposals at once, f(X[n, ndim]) -> [n], as torch tensors on the GPU
        self.batched = bool(batched)
        self.keep_walkers = max(1, min(int(keep_walkers), self.nwalkers))
        self.seed = int(np.random.SeedSequence(seed).generate_state(1, dtype=np.uint64)[0])
        self.stream = np.random.default_rng(self.seed)      # for host-side custom jumps that want a generator
        self.ndim = ndim
        self.logl_spec = logl if isinstance(logl, tuple) else None
        self.logp_spec = logp if isinstance(logp, tuple) else None
        self.logl = None if self.logl_spec else _function_wrapper(logl, loglargs, loglkwargs)
        self.logp = None if self.logp_spec else _function_wrapper(logp, logpargs, logpkwargs)
        if (self.logl_spec is None) != (self.logp_spec is None):
            raise ValueError("logl and logp must both be callables or both be device specifications")
        self.logl_grad = self.logp_grad = None
        # device likelihoods have analytic gradients built in: logl_grad=True, logp_grad=True turns the gradient jumps on
        self.device_grads = self.logl_spec is not None and logl_grad is True and logp_grad is True
        if self.logl_spec is not None and not self.device_grads and (logl_grad is not None or logp_grad is not None):
            raise ValueError("with a device likelihood pass logl_grad=True, logp_grad=True to use its built-in gradients")
        if self.logl_spec is None and logl_grad is not None and logp_grad is not None:
            self.logl_grad = _function_wrapper(logl_grad, loglargs, loglkwargs)
            self.logp_grad = _function_wrapper(logp_grad, logpargs, logpkwargs)
        self.outDir, self.verbose, self.resume = outDir, verbose, resume
        if not os.path.exists(self.outDir):
            try:
                os.makedirs(self.outDir)
            except OSError:
                pass
        self.groups = groups
        if groups is None:
            self.groups = [np.arange(0, self.ndim)]
        self.cov = cov                                       # kept by reference and updated in place (:134, :794)
        self.U, self.S = [[]] * len(self.groups), [[]] * len(self.groups)
        for ct, group in enumerate(self.groups):             # :139-145
            cg = np.asarray(self.cov, dtype=np.float64)[np.ix_(np.asarray(group), np.asarray(group))]
            self.U[ct], self.S[ct], _ = np.linalg.svd(cg)
        self.M2 = np.zeros((ndim, ndim))
        self.mu = np.zeros(ndim)
        self.propCycle, self.jumpDict, self.aux = [], {}, []
        self.engine = None
        self._ctx = (0, 0)

    # ------------------------------------------------------------------ proposal cycle API
    def addProposalToCycle(self, func, weight):
        """PTMCMCSampler.py:988-1014: ``weight`` copies of ``func`` join the cycle; weight 0 is ignored."""
        if weight == 0:
            return
        for _ in range(weight):
            self.propCycle.append(func)
        if func.__name__ not in self.jumpDict:
            self.jumpDict[func.__name__] = [0, 0]
            open(self.outDir + "/" + func.__name__ + "_jump.txt", "w").close()

    def addAuxilaryJump(self, func):
        """PTMCMCSampler.py:1017-1028."""
        self.aux.append(func)

    def randomizeProposalCycle(self):
        """PTMCMCSampler.py:1031-1045 (the shuffled copy is dead state there too)."""
        index = np.arange(len(self.propCycle))
        self.stream.shuffle(index)
        self.randomizedPropCycle = [self.propCycle[ind] for ind in index]

    # the three built-in jumps exist as named cycle entries; their arithmetic runs on the device
    def covarianceJumpProposalSCAM(self, x, iter, beta):
        raise RuntimeError("built-in jump: evaluated inside the HIP kernels")

    def covarianceJumpProposalAM(self, x, iter, beta):
        raise RuntimeError("built-in jump: evaluated inside the HIP kernels")

    def DEJump(self, x, iter, beta):
        raise RuntimeError("built-in jump: evaluated inside the HIP kernels")

    def _builtin(self, f):
        for k, b in enumerate((self.covarianceJumpProposalSCAM, self.covarianceJumpProposalAM, self.DEJump)):
            if f == b:
                return k
        return -1

    def temperatureLadder(self, Tmin, Tmax=None, tstep=None):
        return temperature_ladder(self.nchain, self.ndim, Tmin, Tmax, tstep)

    # ------------------------------------------------------------------ initialize (:157-319)
    def initialize(self, Niter, ladder=None, Tmin=1, Tmax=None, Tskip=100, isave=1000, covUpdate=1000, SCAMweight=30,
                   AMweight=20, DEweight=50, NUTSweight=20, HMCweight=20, MALAweight=0, burn=50000, HMCstepsize=0.1,
                   HMCsteps=300, maxIter=None, thin=10, i0=0, neff=None, writeHotChains=False, hotChain=False):
        from .engine import PTEngine
        if maxIter is None:
            maxIter = Niter
        self.ladder, self.covUpdate, self.burn, self.Tskip = ladder, covUpdate, burn, Tskip
        self.SCAMweight, self.AMweight, self.DEweight = SCAMweight, AMweight, DEweight
        self.thin, self.isave, self.Niter, self.neff, self.tstart = thin, isave, Niter, neff, 0
        N = int(maxIter / thin) + 1
        kw = self.keep_walkers
        self._chains = np.zeros((kw, N, self.ndim))
        self._lnlikes, self._lnprobs = np.zeros((kw, N)), np.zeros((kw, N))
        self._chain, self._lnlike, self._lnprob = self._chains[0], self._lnlikes[0], self._lnprobs[0]
        self.ind_next_write = 0
        self.naccepted = self.swapProposed = self.nswap_accepted = 0
        if self.logl_grad is not None and self.logp_grad is not None:                  # :226-258, same order
            from .gradjump import HMCJump, NUTSJump
            lg, pg, cov, nb = self.logl_grad, self.logp_grad, self.cov, self.burn
            if MALAweight > 0 and self.verbose:                                        # :229-235
                print("WARNING: MALA jumps are not provided (the reference flags them as not working properly, "
                      "PTMCMCSampler.py:230-231): MALAweight ignored")
            if HMCweight > 0:
                self.addProposalToCycle(_PerRankJump(self, lambda: HMCJump(lg, pg, cov, nb, stepsize=HMCstepsize, nminsteps=2,
                                                                          nmaxsteps=HMCsteps)), HMCweight)
            if NUTSweight > 0:
                self.addProposalToCycle(_PerRankJump(self, lambda: NUTSJump(lg, pg, cov, nb, trajectoryDir=None,
                                                                           write_burnin=False, force_trajlen=None,
                                                                           force_epsilon=None, delta=0.6)), NUTSweight)
        self._grad_weights = (0, 0)
        if self.device_grads:                                                          # :226-258 on the device (csrc/ptmi_gj.inc.h)
            if MALAweight > 0 and self.verbose:
                print("WARNING: MALAJump is not built for the device likelihoods (the reference flags it as not working, "
                      "PTMCMCSampler.py:230-231): MALAweight ignored")
            self._grad_weights = (int(NUTSweight), int(HMCweight))
            for name, wgt in (("HMCJump", HMCweight), ("NUTSJUMP", NUTSweight)):       # the reference's jump names
                if wgt > 0:
                    self.jumpDict[name] = [0, 0]
                    open(self.outDir + "/" + name + "_jump.txt", "w").close()
        self.addProposalToCycle(self.covarianceJumpProposalSCAM, self.SCAMweight)     # :261
        self.addProposalToCycle(self.covarianceJumpProposalAM, self.AMweight)         # :264
        if len(self.propCycle) == 0 and sum(self._grad_weights) == 0:
            raise ValueError("No jump proposals specified!")
        self.randomizeProposalCycle()
        if self.ladder is None:
            self.ladder = self.temperatureLadder(Tmin, Tmax=Tmax)
        self.temp = self.ladder[0]
        self.fname = self.outDir + "/chain_{0}.txt".format(self.temp)                # :285
        self.writeHotChains, self.hotChain = writeHotChains, hotChain
        # the hotter ranks of walker 0 (the other MPI ranks' own chain files in the reference, :285-288, :346)
        self._hot_names = []
        if writeHotChains and self.nchain > 1:
            self._hot = np.zeros((self.nchain - 1, N, self.ndim))
            self._hot_lnl, self._hot_lnp = np.zeros((self.nchain - 1, N)), np.zeros((self.nchain - 1, N))
            for r in range(1, self.nchain):
                last_hot = hotChain and r == self.nchain - 1
                self._hot_names.append(self.outDir + ("/chain_hot.txt" if last_hot else "/chain_{0}.txt".format(self.ladder[r])))
        self.resumeLength = 0
        self._ckpt = os.path.join(self.outDir, "ptmi_checkpoint.npz")
        have_file = bool(self.resume) and os.path.isfile(self.fname)
        self._resuming = have_file and os.path.isfile(self._ckpt)          # continue from the device checkpoint
        self._replaying = have_file and not self._resuming                 # the reference's way: replay the chain file
        self.resumechain = None
        if self._resuming or self._replaying:
            if self.verbose:
                print("Resuming run from chain file {0}".format(self.fname))
        if self._replaying:
            # PTMCMCSampler.py:290-313: the text rows are all there is (chains the reference wrote, or a run of ours without
            # checkpoints).  In the reference every MPI rank replays its own chain_<T>.txt; here the one process replays
            # all of a ladder's files (one walker: a file holds one rank of one reference run).
            rank_files = [self.fname]
            for r in range(1, self.nchain):
                last_hot = hotChain and r == self.nchain - 1
                rank_files.append(self.outDir + ("/chain_hot.txt" if last_hot else "/chain_{0}.txt".format(self.ladder[r])))
            # several walkers of ONE temperature: every walker is a run of its own with its own file (chain_1.txt, chain_1_w<k>.txt,
            # written when keep_walkers == nwalkers), replayed side by side
            walker_files = [self.fname if k == 0 else self.fname[:-4] + "_w%d.txt" % k for k in range(self.nwalkers)]
            batch_ok = self.nwalkers > 1 and self.nchain == 1 and self.keep_walkers == self.nwalkers
            if batch_ok:
                rank_files = walker_files
            missing = [f for f in rank_files if not os.path.isfile(f)]
            if (self.nwalkers != 1 and not batch_ok) or missing:
                raise Exception("Couldn't resume: {0} exists but the device checkpoint {1} does not, and chain files alone can only be "
                                "replayed for one walker with the file of every temperature present, or for the walkers of one "
                                "temperature with a file each (keep_walkers = nwalkers) ({2}): other batches are resumable when "
                                "they were started with checkpoint=True (or resume=True).  Refusing to "
                                "overwrite it.".format(self.fname, self._ckpt, "missing: " + ", ".join(missing) if missing else
                                                       "nwalkers = %d, ntemps = %d, keep_walkers = %d" % (self.nwalkers, self.nchain, self.keep_walkers)))
            try:
                self._resume_rows = [np.loadtxt(f, ndmin=2) for f in rank_files]
            except ValueError as error:
                print("Reading old chain files failed with error", error)
                raise Exception("Couldn't read old chain to resume")
            self._resume_by_walker = batch_ok
            self.resumechain = self._resume_rows[0]
            self.resumeLength = self.resumechain.shape[0]
            for f, rows in zip(rank_files, self._resume_rows):
                if rows.shape[1] != self.ndim + 4:
                    raise Exception("Old chain {0} has {1} columns, expected ndim + 4 = {2}".format(f, rows.shape[1], self.ndim + 4))
                if rows.shape[0] != self.resumeLength:
                    raise Exception("Old chains differ in length: {0} has {1} rows, {2} has {3}".format(f, rows.shape[0], self.fname,
                                                                                                      self.resumeLength))
            if self.isave != self.thin and self.resumeLength % (self.isave / self.thin) != 1:
                raise Exception("Old chain has {0} rows, which is not the initial sample plus a multiple of isave/thin = {1}".format(
                    self.resumeLength, self.isave // self.thin))
            print("Resuming with", self.resumeLength, "samples from file representing", (self.resumeLength - 1) * self.thin + 1,
                  "original samples")
        if not (self._resuming or self._replaying):
            open(self.fname, "w").close()
            for f in self._hot_names:
                open(f, "w").close()
            # a fresh start owns the directory: a checkpoint of an earlier run must not survive beside the new chain file
            # (a later resume=True would continue THAT run and cut the newer chain file to its row count)
            for stale in (self._ckpt, self._ckpt + ".tmp.npz"):
                if os.path.isfile(stale):
                    os.remove(stale)
        # ---- engine
        self.host_jumps = [f for f in self.propCycle if self._builtin(f) < 0]
        self.split = self.logl is not None or bool(self.host_jumps) or bool(self.aux)
        if self.split and self.logl is None:
            raise NotImplementedError("host-side jumps need Python logl/logp callbacks")
        if self.batched and (self.logl is None or self.host_jumps or self.aux):
            raise NotImplementedError("batched=True takes callable logl/logp and no per-chain Python jumps "
                                      "(they would need every proposal on the host)")
        self.engine = PTEngine(
            self.ndim, self.nchain, self.nwalkers, np.asarray(self.cov, dtype=np.float64), ladder=self.ladder,
            logl=self.logl_spec or ("iso",), logp=self.logp_spec or ("flat",),
            weights=(self.SCAMweight, self.AMweight, self.DEweight), cov_update=covUpdate, burn=burn, tskip=Tskip,
            seed=self.seed, cov_mode=self.cov_mode, hot_chain=hotChain, device=self.device_index, split=self.split,
            swap_mode=self.swap_mode, pick_mode=self.pick_mode, eig_mode=self.eig_mode, grad_weights=self._grad_weights, hmc=(HMCstepsize, 2, HMCsteps), nuts_maxdepth=self.nuts_maxdepth,
            w_host=len(self.host_jumps), keep_lnl=True, groups=None if len(self.groups) == 1 and len(self.groups[0]) == self.ndim and np.array_equal(np.asarray(self.groups[0]), np.arange(self.ndim)) else self.groups)

    # ------------------------------------------------------------------ sample (:374-528)
    def sample(self, p0, Niter, ladder=None, Tmin=1, Tmax=None, Tskip=100, isave=1000, covUpdate=1000, SCAMweight=20,
               AMweight=20, DEweight=20, NUTSweight=20, MALAweight=20, HMCweight=20, burn=10000, HMCstepsize=0.1,
               HMCsteps=300, maxIter=None, thin=10, i0=0, neff=None, writeHotChains=False, hotChain=False):
        if maxIter is None:
            maxIter = Niter
        if isave % thin != 0:
            raise ValueError("isave = %d is not a multiple of thin =  %d" % (isave, thin))
        if Niter % thin != 0:
            print("Niter = %d is not a multiple of thin = %d.  The last %d samples will be lost" % (Niter, thin, Niter % thin))
        if self.logl_grad is None and not self.device_grads:
            NUTSweight = MALAweight = HMCweight = 0
        if i0 == 0:
            self.initialize(Niter, ladder=ladder, Tmin=Tmin, Tmax=Tmax, Tskip=Tskip, isave=isave, covUpdate=covUpdate,
                            SCAMweight=SCAMweight, AMweight=AMweight, DEweight=DEweight, NUTSweight=NUTSweight,
                            MALAweight=MALAweight, HMCweight=HMCweight, burn=burn, HMCstepsize=HMCstepsize,
                            HMCsteps=HMCsteps, maxIter=maxIter, thin=thin, i0=i0, neff=neff,
                            writeHotChains=writeHotChains, hotChain=hotChain)
        eng = self.engine
        if eng is None:
            raise RuntimeError("sample(..., i0 != 0) continues an initialised sampler: call sample() with i0 = 0 first")
        p0 = np.asarray(p0, dtype=np.float64)
        self.tstart = time.time()
        if i0 != 0:
            self.Niter = Niter                                 # the chains take p0 as their state at iteration i0 (:479-491)
        if i0 == 0 and self._resuming:
            i0 = self._load_checkpoint()                       # continue where the last complete save stopped
        elif i0 == 0 and self._replaying:
            i0 = self._replay_chain_file()                     # rebuild the adaptive state from the rows, as the reference does
        else:
            if self.split:
                self._init_split(p0, i0)
            else:
                eng.init_state(p0, i0)
            self._harvest([i0])
            if self._hot_names:
                self._harvest_hot(i0)
            if i0 % self.isave == 0:
                self.writeOutput(i0)
        it = i0 + 1
        message = "\nRun Complete"
        while it <= self.Niter:
            before = eng.eig_epochs
            eng._epochs(it)
            if eng.eig_epochs != before:
                self._mirror_cov()
            if (it - 1) == self.burn and self.DEweight and self.DEJump not in self.propCycle:     # :579-585
                if self.verbose:
                    print("Adding DE jump with weight {0}".format(self.DEweight))
                self.addProposalToCycle(self.DEJump, self.DEweight)
                self.randomizeProposalCycle()
            if self.split and self.batched:
                # batched device callbacks: a whole segment (no epoch, swap, save or hot-rank sample inside), the accept test of an
                # iteration and the next proposal in one launch (PTEngine.callback_segment)
                end = min(eng._segment_end(it, self.Niter), ((it - 1) // self.isave + 1) * self.isave)
                if self._hot_names:
                    end = min(end, ((it - 1) // self.thin + 1) * self.thin)
                eng.callback_segment(it, end, self.logl, self.logp)
            elif self.split:
                end = it
                self._split_step(it)
            else:
                end = min(eng._segment_end(it, self.Niter), ((it - 1) // self.isave + 1) * self.isave)
                if self._hot_names:                                  # hot ranks are sampled from the state, at thin multiples
                    end = min(end, ((it - 1) // self.thin + 1) * self.thin)
                eng.mh_steps(it, end - it + 1)
            if self.Tskip > 0 and self.nchain > 1 and end % self.Tskip == 0:
                eng.swap(end)
            self._harvest([i for i in range(it, end + 1) if i % self.thin == 0])
            if self._hot_names and end % self.thin == 0:
                self._harvest_hot(end)
            if end % self.isave == 0:
                self.writeOutput(end)
            if self.neff and end % 1000 == 0 and end > 2 * self.burn:                               # :510-521
                # acor.acor(chain[burn:iter-1, ii])[0] per dimension, Neff = iter / max(1, nanmax(tau)): with thin = 1 exactly the
                # reference's expression (ess.acor restates the un-vendored package's published algorithm).  With thin > 1 the
                # reference's slice runs past the rows written so far; here the rows [burn / thin, iter / thin) that exist are used.
                from .ess import AcorError, acor
                lo, hi = (self.burn, end - 1) if self.thin == 1 else (self.burn // self.thin, end // self.thin)
                taus = []
                for ii in range(self.ndim):
                    try:
                        taus.append(acor(self._chain[lo:hi, ii])[0])
                    except AcorError as err:                         # "autocorrelation time too long": the reference's run would die
                        taus = []                                    # here with acor's RuntimeError; this one keeps sampling and asks again
                        if self.verbose:                             # in 1000 iterations -- and says so
                            print("neff check skipped at iteration {0}: {1} (parameter {2}, {3} rows)".format(end, err, ii, hi - lo))
                        break
                if len(taus) and np.isfinite(taus).any():
                    Neff = (end // self.thin) / max(1.0, np.nanmax(taus))
                    if int(Neff) >= self.neff:
                        message = "\nRun Complete with {0} effective samples".format(int(Neff))
                        self.Niter = end
            it = end + 1
        eng.iter = self.Niter
        self.writeOutput(self.Niter)
        if self.verbose:
            print(message)

    # ------------------------------------------------------------------ checkpoint / resume
    # The reference resumes by replaying its text chain file (PTMCMCSampler.py:290-319, 591-599): the adaptive
    # state and the RNG are not restored there.  Here every save also writes the device state; with the
    # counter-based RNG a resumed run continues bit-identically to an uninterrupted one (host-side jump objects --
    # custom, HMC, NUTS -- keep their own state and are not part of the checkpoint).
    def _save_checkpoint(self, iter):
        st = self.engine.checkpoint()
        st["iter"] = iter
        st["f_format"] = _CKPT_FORMAT
        st["f_fingerprint"] = self._fingerprint()
        n = self.ind_next_write                                # the stored part of the sample arrays only
        st.update(f_chains=self._chains[:, :n], f_lnlikes=self._lnlikes[:, :n], f_lnprobs=self._lnprobs[:, :n],
                  f_ind_next_write=self.ind_next_write,
                  f_jump_names=np.asarray(list(self.jumpDict), dtype=str),
                  f_jump_counts=np.asarray([self.jumpDict[k] for k in self.jumpDict], dtype=np.int64).reshape(-1, 2),
                  f_de_in_cycle=int(self.DEJump in self.propCycle))
        tmp = self._ckpt + ".tmp.npz"
        np.savez(tmp, **st)
        os.replace(tmp, self._ckpt)

    _FP_NAMES = ("seed", "ndim", "ntemps", "nwalkers", "thin", "covUpdate", "burn", "DE row stride", "DE row format", "keep_walkers",
                 "AM row format", "ladder", "Tskip", "jump weights", "engine modes", "likelihood / prior")

    def _fingerprint(self):
        """What a checkpoint must agree on with the run that loads it: sizes and formats, and (as 63-bit digests) the ladder,
        the jump weights, the engine modes and the likelihood / prior specification -- a resumed run that differs in any of
        them would continue the old device state under a different sampler."""
        import hashlib
        eng = self.engine

        def digest(*parts):
            h = hashlib.sha256()
            for p in parts:
                h.update(np.ascontiguousarray(p).tobytes() if isinstance(p, np.ndarray) else repr(p).encode())
            return int.from_bytes(h.digest()[:8], "little") & 0x7FFFFFFFFFFFFFFF

        def spec(s):        # a device specification ("dense", mu, P) by value; Python callbacks by name only
            if s is None:
                return ("callback",)
            return tuple(np.asarray(v, dtype=np.float64) if not isinstance(v, str) else v for v in s)

        groups = [np.asarray(g, dtype=np.int64) for g in self.groups]
        return np.asarray([self.seed & 0x7FFFFFFFFFFFFFFF, self.ndim, self.nchain, self.nwalkers, self.thin, self.covUpdate, self.burn,
                           eng.de_ld, eng.de_epl, self.keep_walkers, eng.am_epl,
                           digest(np.asarray(eng.ladder, dtype=np.float64), np.asarray(eng.temps_mh, dtype=np.float64)), self.Tskip,
                           digest((self.SCAMweight, self.AMweight, self.DEweight) + tuple(self._grad_weights) + (len(self.host_jumps),)),
                           digest((self.cov_mode, self.swap_mode, self.pick_mode, self.eig_mode, self.nuts_maxdepth, bool(self.split), bool(self.batched),
                                   bool(eng.am_rle)), *groups),
                           digest(*(spec(self.logl_spec) + spec(self.logp_spec)))], dtype=np.int64)

    def _load_checkpoint(self):
        st = np.load(self._ckpt, allow_pickle=False)
        if "f_format" not in st.files or int(st["f_format"]) != _CKPT_FORMAT:
            was = int(st["f_format"]) if "f_format" in st.files else 1
            why = ", ".join(_CKPT_FORMAT_WHY[v] for v in range(max(was, 1) + 1, _CKPT_FORMAT + 1) if v in _CKPT_FORMAT_WHY)
            raise Exception("{0} was written in checkpoint format {1}; this build reads format {2} (changed since: {3}): it cannot be "
                            "resumed from".format(self._ckpt, was, _CKPT_FORMAT, why or "unknown"))
        fp = self._fingerprint()
        if not np.array_equal(st["f_fingerprint"], fp):
            bad = [n for n, a, b in zip(self._FP_NAMES, st["f_fingerprint"], fp) if a != b] or ["fingerprint length"]
            raise Exception("{0} belongs to a different run ({1} differ): refusing to resume from it".format(self._ckpt, ", ".join(bad)))
        self.engine.restore(st)
        n = min(self._chains.shape[1], st["f_chains"].shape[1])
        self._chains[:, :n], self._lnlikes[:, :n], self._lnprobs[:, :n] = st["f_chains"][:, :n], st["f_lnlikes"][:, :n], st["f_lnprobs"][:, :n]
        self.ind_next_write = int(st["f_ind_next_write"])
        if int(st["f_de_in_cycle"]) and self.DEJump not in self.propCycle:
            self.addProposalToCycle(self.DEJump, self.DEweight)
        for name, cnt in zip(st["f_jump_names"], st["f_jump_counts"]):
            self.jumpDict[str(name)] = [int(cnt[0]), int(cnt[1])]
        self._mirror_cov()
        self.resumeLength = self.ind_next_write
        # the chain files must end where the checkpoint does: a run killed between the file write and the checkpoint
        # leaves rows the resumed run is about to write again
        names = [self.fname if k == 0 else self.fname[:-4] + "_w%d.txt" % k for k in range(self.keep_walkers)] + list(self._hot_names)
        for fname in names:
            if not os.path.isfile(fname):
                continue
            with open(fname, "rb+") as fh:                        # streamed: find the byte offset of row ind_next_write
                nrows, cut = 0, None
                for line in iter(fh.readline, b""):
                    nrows += 1
                    if nrows == self.ind_next_write:
                        cut = fh.tell()
                        break
                if nrows < self.ind_next_write:
                    raise Exception("{0} has {1} rows but the checkpoint was written after {2}".format(fname, nrows, self.ind_next_write))
                fh.truncate(cut if self.ind_next_write > 0 else 0)
        i0 = int(st["iter"])
        print("Resuming with", self.resumeLength, "samples from file representing", i0 + 1, "original samples")
        return i0

    def _replay_chain_file(self):
        """Resume as the reference does (PTMCMCSampler.py:591-599): for iterations below resumeLength * thin a chain does
        not jump but takes row iter // thin of its old file as its state, so the AM buffer, the covariance epochs
        (:545-560) and the DE history (:563-571) are rebuilt from the files.  With a ladder every rank replays its own
        file and the swaps of the replayed iterations still run (:624-627 follows the replay branch): PTswap acts on the
        replayed states, its result is what the AM buffer of that iteration holds and what the swap counters count, and the
        next iteration takes the files' rows again.  Returns the last replayed iteration."""
        import torch
        eng, thin, cu = self.engine, self.thin, self.covUpdate
        d, n = self.ndim, self.nchain
        last = self.resumeLength * thin - 1                         # iterations 1 .. last are replayed
        if getattr(self, "_resume_by_walker", False):
            return self._replay_walker_files(last)
        betas = 1.0 / eng.temps_mh
        Xs = [rows[:, :d] for rows in self._resume_rows]
        lnls = [rows[:, -3] for rows in self._resume_rows]
        lnps = [rows[:, -4] for rows in self._resume_rows]
        with np.errstate(invalid="ignore"):                        # log-prior of a row (a -inf row: lnlike may be -inf too)
            lprs = [np.where(np.isneginf(lnps[r]), -np.inf, lnps[r] - betas[r] * lnls[r]) for r in range(n)]
        X, lnl, lnp, lpr = Xs[0], lnls[0], lnps[0], lprs[0]

        def set_state(k):
            """every rank holds row k of its file (by rank: a swap may have permuted the slots)"""
            so = eng.get("slot_of")[0].astype(np.int64)
            state, sl, sp = np.zeros((1, n, d)), np.zeros((1, n)), np.zeros((1, n))
            for r in range(n):
                state[0, so[r]], sl[0, so[r]], sp[0, so[r]] = Xs[r][k], lnls[r][k], lprs[r][k]
            eng.t["X"].copy_(torch.from_numpy(state))
            eng.put("lnL", sl)
            eng.put("lp", sp)

        # iteration 0: the first row (:474-476, :491)
        eng.t["AM"][0, 0] = torch.from_numpy(eng.am_rows(X[0]).copy())
        if eng.am_rle:                                              # replayed rows are stored rows: KEY (include/ptmi.h AMflag)
            eng.t["AMflag"].fill_(_lib.AMROW_KEY)
        eng.t["AMaux"][0, 0, 0], eng.t["AMaux"][0, 0, 1] = float(lnl[0]), float(lpr[0])
        self._chains[0, 0], self._lnlikes[0, 0], self._lnprobs[0, 0] = X[0], lnl[0], lnp[0]
        swapped_last = False
        it = 1
        while it <= last:
            before = eng.eig_epochs
            eng._epochs(it)                                         # covariance / DE epochs see the replayed AM rows
            if eng.eig_epochs != before:
                self._mirror_cov()
            if (it - 1) == self.burn and self.DEweight and self.DEJump not in self.propCycle:
                self.addProposalToCycle(self.DEJump, self.DEweight)
                self.randomizeProposalCycle()
            end = min(eng._segment_end(it, last), last)
            its = np.arange(it, end + 1)
            src = its // thin
            eng.t["AM"][0, torch.from_numpy(its % cu).to(eng.device)] = torch.from_numpy(np.ascontiguousarray(eng.am_rows(X[src]))).to(eng.device)
            aux = np.stack([lnl[src], lpr[src]], 1)
            eng.t["AMaux"][0, torch.from_numpy(its % cu).to(eng.device)] = torch.from_numpy(aux).to(eng.device)
            keep = its[its % thin == 0]
            self._chains[0, keep // thin], self._lnlikes[0, keep // thin], self._lnprobs[0, keep // thin] = (
                X[keep // thin], lnl[keep // thin], lnp[keep // thin])
            swapped_last = False
            if n > 1 and self.Tskip > 0 and end % self.Tskip == 0:   # PTswap of a replayed iteration (:624-627)
                set_state(end // thin)
                eng.swap(end)                                       # also stores the AM row of the state now at rank 0
                swapped_last = end == last
            it = end + 1
        # the chains' states after the replay, their acceptance counters (:597-599), and what is on file already
        k = last // thin
        if not swapped_last:
            set_state(k)
        so = eng.get("slot_of")[0].astype(np.int64)
        nacc = eng.get("nacc")
        for r in range(n):
            nacc[0, r] = int(round(last * self._resume_rows[r][k, -2]))
        eng.put("nacc", nacc.astype(np.int64))
        self.ind_next_write = self.resumeLength
        eng.iter = last
        return last

    def _replay_walker_files(self, last):
        """_replay_chain_file for the walkers of ONE temperature: walker w replays its own file (a reference run each,
        PTMCMCSampler.py:591-599) -- its rows are its AM buffer, the covariance epochs (:545-560; per walker, or pooled over all
        walkers' rows) and the DE history (:563-571) are rebuilt from them -- and all walkers continue together."""
        import torch
        eng, thin, cu, d, W = self.engine, self.thin, self.covUpdate, self.ndim, self.nwalkers
        rows = self._resume_rows
        X = np.stack([r[:, :d] for r in rows])                      # [W][rows][d]
        lnl = np.stack([r[:, -3] for r in rows])
        lnp = np.stack([r[:, -4] for r in rows])
        beta0 = 1.0 / eng.temps_mh[0]
        with np.errstate(invalid="ignore"):
            lpr = np.where(np.isneginf(lnp), -np.inf, lnp - beta0 * lnl)
        dev = eng.device
        eng.t["AM"][:, 0] = torch.from_numpy(np.ascontiguousarray(eng.am_rows(X[:, 0]))).to(dev)
        if eng.am_rle:
            eng.t["AMflag"].fill_(_lib.AMROW_KEY)
        eng.t["AMaux"][:, 0, 0], eng.t["AMaux"][:, 0, 1] = torch.from_numpy(lnl[:, 0]).to(dev), torch.from_numpy(lpr[:, 0]).to(dev)
        self._chains[:, 0], self._lnlikes[:, 0], self._lnprobs[:, 0] = X[:, 0], lnl[:, 0], lnp[:, 0]
        it = 1
        while it <= last:
            before = eng.eig_epochs
            eng._epochs(it)
            if eng.eig_epochs != before:
                self._mirror_cov()
            if (it - 1) == self.burn and self.DEweight and self.DEJump not in self.propCycle:
                self.addProposalToCycle(self.DEJump, self.DEweight)
                self.randomizeProposalCycle()
            end = min(eng._segment_end(it, last), last)
            its = np.arange(it, end + 1)
            src = its // thin
            ring = torch.from_numpy(its % cu).to(dev)
            eng.t["AM"][:, ring] = torch.from_numpy(np.ascontiguousarray(eng.am_rows(X[:, src]))).to(dev)
            eng.t["AMaux"][:, ring] = torch.from_numpy(np.stack([lnl[:, src], lpr[:, src]], 2)).to(dev)
            keep = its[its % thin == 0] // thin
            self._chains[:, keep], self._lnlikes[:, keep], self._lnprobs[:, keep] = X[:, keep], lnl[:, keep], lnp[:, keep]
            it = end + 1
        k = last // thin
        eng.t["X"].copy_(torch.from_numpy(np.ascontiguousarray(X[:, k][:, None, :])))
        eng.put("lnL", lnl[:, k][:, None])
        eng.put("lp", lpr[:, k][:, None])
        nacc = eng.get("nacc")
        for w in range(W):
            nacc[w, 0] = int(round(last * rows[w][k, -2]))           # :597-599
        eng.put("nacc", nacc.astype(np.int64))
        self.ind_next_write = self.resumeLength
        eng.iter = last
        return last

    # ------------------------------------------------------------------ bookkeeping
    def _mirror_cov(self):
        cov = self.engine.get("cov")[0]
        self.cov[:, :] = cov                                      # :794, in place
        Ut, Sv = self.engine.get("Ut")[0], self.engine.get("S")[0]
        for ct, group in enumerate(self.groups):
            g = np.asarray(group)
            self.U[ct] = Ut[ct][np.ix_(np.arange(len(g)), g)].T.copy()
            self.S[ct] = Sv[ct][:len(g)].copy()
        self.M2, self.mu = self.engine.get("M2")[0], self.engine.get("mu")[0]

    def _counters(self):
        eng = self.engine
        self.naccepted = int(eng.get("nacc")[0, 0])
        self.nswap_accepted = int(eng.get("nswap")[0, 0])
        self.swapProposed = eng.swap_proposed
        js = eng.get("jstat")[0, 0]
        for k, f in enumerate((self.covarianceJumpProposalSCAM, self.covarianceJumpProposalAM, self.DEJump)):
            if f.__name__ in self.jumpDict:
                self.jumpDict[f.__name__] = [int(js[k, 0]), int(js[k, 1])]
        for k, name in ((_lib.J_NUTS, "NUTSJUMP"), (_lib.J_HMC, "HMCJump")):
            if self._grad_weights[k - _lib.J_NUTS] > 0:
                self.jumpDict[name] = [int(js[k, 0]), int(js[k, 1])]

    def _harvest(self, iters):
        """updateChains (:331-335) for the kept walkers, read back from the AM ring."""
        if not iters:
            return
        eng, kw = self.engine, self.keep_walkers
        rows = [i % eng.cov_update for i in iters]
        eng.am_expand(0, kw, min(iters), max(iters))                # repeats copied forward for what is read here (a no-op with stored rows)
        X = eng.am_params(eng.t["AM"][:kw][:, rows].cpu().numpy())
        aux = eng.t["AMaux"][:kw][:, rows].cpu().numpy()
        beta0 = 1.0 / eng.temps_mh[0]
        for n, i in enumerate(iters):
            ind = int(i / self.thin)
            if ind >= self._chains.shape[1]:
                continue
            self._chains[:, ind] = X[:, n]
            self._lnlikes[:, ind] = aux[:, n, 0]
            self._lnprobs[:, ind] = beta0 * aux[:, n, 0] + aux[:, n, 1]

    def _harvest_hot(self, i):
        """Current state of walker 0's ranks 1.. (post-swap, like updateChains at :627)."""
        eng = self.engine
        ind = int(i / self.thin)
        if ind >= self._hot.shape[1]:
            return
        rows = eng.t["slot_of"][0, 1:].long()
        self._hot[:, ind] = eng.t["X"][0, rows].cpu().numpy()
        lnl, lp = eng.t["lnL"][0, rows].cpu().numpy(), eng.t["lp"][0, rows].cpu().numpy()
        self._hot_lnl[:, ind] = lnl
        self._hot_lnp[:, ind] = (1.0 / eng.temps_mh[1:]) * lnl + lp

    # ------------------------------------------------------------------ host-callback path
    def _eval_host(self, Q):
        """logp then logl of every proposal, with the reference's short-circuit (:605-611)."""
        lp = np.empty(Q.shape[:2])
        ll = np.zeros(Q.shape[:2])
        for w in range(Q.shape[0]):
            for s in range(Q.shape[1]):
                v = self.logp(Q[w, s])
                lp[w, s] = v
                if v != float(-np.inf):
                    ll[w, s] = self.logl(Q[w, s])
        return ll, lp

    def _init_split(self, p0, i0=0):
        import torch
        eng = self.engine
        if self.batched:
            eng.init_state_callback(p0, self.logl, self.logp, i0)
            return
        full = p0 if p0.ndim == 3 else np.broadcast_to(p0, (eng.W, eng.nt, eng.d)).copy()
        eng.t["X"].copy_(torch.from_numpy(np.ascontiguousarray(full)))
        ll, lp = self._eval_host(full)
        ll[lp == -np.inf] = -np.inf                               # :481-483
        eng.put("lnL", ll)
        eng.put("lp", lp)
        eng._store_initial(i0)
        eng.iter = int(i0)

    def _split_step(self, it):
        import torch
        eng = self.engine
        if self.batched:
            eng.split_step(it, self.logl, self.logp)
            return
        _lib.check(eng.lib.ptmi_propose(eng.h, it))
        Q, qa = eng.t["Q"].cpu().numpy(), eng.t["qaux"].cpu().numpy()
        temp_of = eng.get("temp_of")
        dirty = False
        if self.host_jumps or self.aux:
            X = eng.get("X")
            for w in range(eng.W):
                for s in range(eng.nt):
                    beta = 1.0 / eng.temps_mh[temp_of[w, s]]
                    jt = int(qa[w, s, 1])
                    if jt >= _lib.J_NTYPES:                        # a cycle entry served on the host (:1059)
                        self._ctx = (w, int(temp_of[w, s]))
                        q, qxy = self.host_jumps[jt - _lib.J_NTYPES](X[w, s], it, beta)
                        Q[w, s], qa[w, s, 0] = q, qxy
                        dirty = True
                    for aux in self.aux:                           # :1062-1065
                        q, qxy_aux = aux(X[w, s], Q[w, s], it, beta)
                        Q[w, s] = q
                        qa[w, s, 0] += qxy_aux
                        dirty = True
        if dirty:
            eng.t["Q"].copy_(torch.from_numpy(Q))
            eng.t["qaux"].copy_(torch.from_numpy(qa))
        ll, lp = self._eval_host(Q)
        nl = torch.from_numpy(ll).to(eng.device)
        npr = torch.from_numpy(lp).to(eng.device)
        _lib.check(eng.lib.ptmi_accept(eng.h, it, nl.data_ptr(), npr.data_ptr()))
        if self.host_jumps:
            dec = eng.t["qaux"][0].cpu().numpy()
            s0 = int(eng.get("slot_of")[0, 0])
            jt = int(qa[0, s0, 1])
            if jt >= _lib.J_NTYPES:
                name = self.host_jumps[jt - _lib.J_NTYPES].__name__
                self.jumpDict[name][0] += 1
                self.jumpDict[name][1] += int(dec[s0, 2] > 0.5)

    # ------------------------------------------------------------------ output files (:341-372, :722-766)
    def writeOutput(self, iter):
        if iter // self.thin >= self.ind_next_write:
            self._counters()
            self._writeToFile(iter)
            if iter > 0:
                np.save(self.outDir + "/cov.npy", self.cov)
                self.engine.iter = iter
                if self.checkpoint:
                    self._save_checkpoint(iter)
            if self.verbose:
                if iter > 0:
                    sys.stdout.write("\r")
                percent = iter / self.Niter * 100
                acceptance = self.naccepted / iter if iter > 0 else 0
                elapsed = time.time() - self.tstart
                sys.stdout.write("Finished %2.2f percent in %f s Acceptance rate = %g" % (percent, elapsed, acceptance))
                sys.stdout.flush()

    def _writeToFile(self, iter):
        write_end = iter // self.thin + 1
        for k in range(self.keep_walkers):
            fname = self.fname if k == 0 else self.fname[:-4] + "_w%d.txt" % k
            with open(fname, "a+") as fh:
                for ind in range(self.ind_next_write, write_end):
                    pt_acc = 1
                    if self.nchain > 1 and self.swapProposed != 0:
                        pt_acc = self.nswap_accepted / self.swapProposed
                    fh.write("\t".join(["%22.22f" % (self._chains[k, ind, kk]) for kk in range(self.ndim)]))
                    fh.write("\t%f\t%f\t%f\t%f\n" % (self._lnprobs[k, ind], self._lnlikes[k, ind],
                                                     self.naccepted / iter if iter > 0 else 0, pt_acc))
        if self._hot_names:
            nacc, nsw = self.engine.get("nacc")[0], self.engine.get("nswap")[0]
            for r, fname in enumerate(self._hot_names, start=1):
                with open(fname, "a+") as fh:
                    for ind in range(self.ind_next_write, write_end):
                        pt_acc = 1
                        if r < self.nchain - 1 and self.swapProposed != 0:
                            pt_acc = int(nsw[r]) / self.swapProposed
                        fh.write("\t".join(["%22.22f" % (self._hot[r - 1, ind, kk]) for kk in range(self.ndim)]))
                        fh.write("\t%f\t%f\t%f\t%f\n" % (self._hot_lnp[r - 1, ind], self._hot_lnl[r - 1, ind],
                                                         int(nacc[r]) / iter if iter > 0 else 0, pt_acc))
        self.ind_next_write = write_end
        with open(self.outDir + "/jumps.txt", "w") as fout:
            njumps = len(self.propCycle)
            seen = []
            for jump in self.propCycle:
                if jump not in seen:
                    seen.append(jump)
            for jump in seen:
                fout.write("%s %4.2g\n" % (jump.__name__, sum(1 for f in self.propCycle if f == jump) / njumps))
        for jump in self.jumpDict:
            with open(self.outDir + "/" + jump + "_jump.txt", "a+") as fout:
                fout.write("%g\n" % (self.jumpDict[jump][1] / max(1, self.jumpDict[jump][0])))

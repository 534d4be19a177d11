"""Device sessions: one per (experiment, xdim).  Keeps the spike-count tensor, the modes and the
posterior blocks resident in HBM between E-step and M-step calls, and hands lazy views back to
callers that still want the reference's list-of-arrays ``infRes``.

Trial sharding (multi-GPU): when the process was launched one-rank-per-GPU (RANK / WORLD_SIZE /
LOCAL_RANK in the environment, as torch.distributed.run sets them) every rank holds the full
count tensor (uint8, small) and processes a contiguous slice of each trial list; M-step sufficient
statistics are summed with an RCCL all-reduce inside the C-ABI.
"""
import os
import time
import weakref

import numpy as np

from . import _hip


# ----------------------------------------------------------------------------------------------
# process-wide communicator description (rank, world size, unique id exchange)
# ----------------------------------------------------------------------------------------------
class _stdout_to_stderr:
    """RCCL prints a version banner on the process's stdout (file descriptor 1) when the first communicator comes
    up; callers that emit machine-readable output on stdout (bench.py: one JSON line) must not see it there."""

    def __enter__(self):
        import sys
        sys.stdout.flush()
        self._saved = os.dup(1)
        os.dup2(2, 1)
        return self

    def __exit__(self, *exc):
        os.dup2(self._saved, 1)
        os.close(self._saved)
        return False


_PROCESS_START = time.time()


class phase_deadline:
    """Watchdog around a start-up phase that can block for ever inside the library (ncclCommInitRank and the first collective have no
    timeout: one rank that never joins holds all the others).  A helper thread waits for the phase to end; when the deadline passes
    first it writes one line to stderr and ends THIS process with a non-zero status (os._exit: the main thread is stuck in a native
    call and cannot be interrupted; nothing is re-executed) - the launcher then sees a failed rank with a message instead of a job that
    hangs until somebody's outer limit kills it without one."""

    def __init__(self, what, seconds, rank=0, exit_code=13):
        self.what, self.seconds, self.rank, self.exit_code = what, float(seconds), rank, exit_code
        import threading
        self._done = threading.Event()
        self._thread = threading.Thread(target=self._watch, daemon=True)

    def _watch(self):
        if not self._done.wait(self.seconds):
            import sys
            try:
                sys.stderr.write('pgpfa: rank %d: %s did not finish within %.0f s (another rank never joined, or RCCL could not reach it); '
                                 'exiting with status %d\n' % (self.rank, self.what, self.seconds, self.exit_code))
                sys.stderr.flush()
            finally:
                os._exit(self.exit_code)

    def __enter__(self):
        self._thread.start()
        return self

    def __exit__(self, *exc):
        self._done.set()
        return False


class _World:
    def __init__(self):
        self.rank = int(os.environ.get('RANK', '0'))
        self.size = int(os.environ.get('WORLD_SIZE', '1'))
        self.local_rank = int(os.environ.get('LOCAL_RANK', str(self.rank)))
        # PGPFA_FORCE_COMM=1 builds the RCCL communicator even for a single rank (exercises the same code path)
        self.enabled = (self.size > 1 or os.environ.get('PGPFA_FORCE_COMM', '0') == '1') and \
            os.environ.get('PGPFA_DISABLE_COMM', '0') != '1'
        self._serial = 0
        if self.enabled:
            # single-node job: keep RCCL's bootstrap on loopback and off InfiniBand probing (both can stall in
            # containers without a routable interface); explicit user settings win
            os.environ.setdefault('NCCL_SOCKET_IFNAME', 'lo')
            os.environ.setdefault('NCCL_IB_DISABLE', '1')
            os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')

    def device(self):
        return self.local_rank if self.enabled else int(os.environ.get('PGPFA_DEVICE', '0'))

    def exchange_unique_id(self):
        """Rendezvous of this single-node job through files; returns (RCCL unique id, path of the id file).

        Rank 0 removes whatever an earlier attempt left under the same key, creates the unique id and publishes it; every
        other rank reads it and acknowledges WITH ITS CONTENT; rank 0 waits until every acknowledgement carries the id it
        published (an acknowledgement of an older id is not one) and only then publishes the go-ahead, again with the id.
        A non-zero rank never trusts a single read: it keeps re-reading the id file, acknowledges again when the id
        changed under it (it had read a leftover before rank 0 replaced it), and leaves only when the go-ahead carries the
        id it holds - a leftover go-ahead of another attempt is waited out, not an error.  Nobody enters ncclCommInitRank
        (which blocks without a timeout until every rank has joined) before every rank is known to be alive and to hold
        the same id: a rank that died at start-up (bad device, out of memory while creating its context) or was never
        started makes every other rank raise after PGPFA_RDZV_TIMEOUT seconds (default 300) - a non-zero exit of the whole
        job instead of a hang.  File names are keyed by MASTER_PORT, the launcher's run id, its restart count
        (TORCHELASTIC_RESTART_COUNT: a restarted worker group is another attempt) and pid (shared parent of all ranks)
        and a per-process serial.  A file older than this process by more than the timeout cannot belong to this attempt
        (rank 0 gives up that long after publishing) and is ignored.
        Every rank's acknowledgement also carries a random 16-byte nonce of ITS process, and the go-ahead repeats the nonces of
        all ranks next to the id: a non-zero rank leaves only on a go-ahead that holds its own nonce, so the complete leftovers
        of a crashed earlier run under the same key (id file + matching go-ahead: the key repeats when the same shell reruns
        the job on the same port) cannot send it into ncclCommInitRank with a dead id before rank 0 has cleaned up."""
        self._serial += 1
        base = os.environ.get('PGPFA_RDZV_DIR', '/tmp')
        key = 'pgpfa_uid_%s_%s_%s_%d_%d' % (os.environ.get('MASTER_PORT', '0'), os.environ.get('TORCHELASTIC_RUN_ID', 'none'),
                                           os.environ.get('TORCHELASTIC_RESTART_COUNT', '0'), os.getppid(), self._serial)
        path = os.path.join(base, key)
        timeout = float(os.environ.get('PGPFA_RDZV_TIMEOUT', '300'))
        deadline = time.time() + timeout

        def publish(name, payload):
            tmp = name + '.tmp.%d' % os.getpid()
            with open(tmp, 'wb') as fh:
                fh.write(payload)
            os.replace(tmp, name)

        def read_fresh(name):
            """Content of a file that can belong to this attempt, else None."""
            try:
                if os.path.getmtime(name) < _PROCESS_START - timeout - 5.0:
                    return None
                with open(name, 'rb') as fh:
                    return fh.read()
            except OSError:
                return None

        def timed_out(what, names):
            return _hip.HipBackendError('rendezvous of rank %d timed out after %.0f s waiting for %s (%s): a rank died at start-up or was '
                                        'never launched' % (self.rank, timeout, what, ', '.join(os.path.basename(m) for m in names)))
        acks = [path + '.ack.%d' % r for r in range(1, self.size)]
        nonce = os.urandom(16)
        if self.rank == 0:
            for stale in [path, path + '.go'] + acks:
                try:
                    os.remove(stale)
                except OSError:
                    pass
            uid = _hip.comm_unique_id()
            publish(path, uid)
            while True:
                got = [read_fresh(nm) for nm in acks]
                missing = [nm for nm, a in zip(acks, got) if a is None or len(a) != 144 or a[:128] != uid]
                if not missing:
                    break
                if time.time() >= deadline:
                    try:
                        os.remove(path)
                    except OSError:
                        pass
                    raise timed_out('the acknowledgement of every rank', missing)
                time.sleep(0.01)
            publish(path + '.go', uid + b''.join(a[128:] for a in got))
            for nm in acks:
                try:
                    os.remove(nm)
                except OSError:
                    pass
            return uid, path
        held = None
        while True:
            uid = read_fresh(path)
            if uid is not None and len(uid) == 128:
                if uid != held:
                    publish(path + '.ack.%d' % self.rank, uid + nonce)
                    held = uid
                go = read_fresh(path + '.go')
                if go is not None and len(go) == 128 + 16 * (self.size - 1) and go[:128] == held and \
                        go[128 + 16 * (self.rank - 1):128 + 16 * self.rank] == nonce:
                    # the go-ahead is written after every rank acknowledged THIS id, and it names this process: not a leftover
                    return held, path
            if time.time() >= deadline:
                raise timed_out("rank 0's unique id" if held is None else "rank 0's go-ahead", [path if held is None else path + '.go'])
            time.sleep(0.01)


WORLD = _World()


def cleanup_rendezvous(path):
    """After the first collective (which every rank has passed) rank 0 removes the rendezvous files."""
    if WORLD.rank == 0:
        for name in (path, path + '.go'):
            try:
                os.remove(name)
            except OSError:
                pass


def shard_slice(n_items, rank, size):
    """Contiguous block partition of range(n_items): the first (n_items % size) ranks get one more."""
    base, rem = divmod(n_items, size)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


# ----------------------------------------------------------------------------------------------
# lazy, device-backed sequences (duck-type the reference's per-trial lists)
# ----------------------------------------------------------------------------------------------
class LazyTrialList:
    """Behaves like the reference's list of per-trial arrays; entry i is fetched from HBM on first use."""

    def __init__(self, n, fetch):
        self._n = n
        self._fetch = fetch
        self._cache = {}

    def __len__(self):
        return self._n

    def __getitem__(self, i):
        if isinstance(i, slice):
            return [self[j] for j in range(*i.indices(self._n))]
        if i < 0:
            i += self._n
        if not 0 <= i < self._n:
            raise IndexError(i)
        if i not in self._cache:
            self._cache[i] = self._fetch(i)
        return self._cache[i]

    def __iter__(self):
        return (self[i] for i in range(self._n))


class DeviceInfRes(dict):
    """infRes dict of inference.laplace / dualVariational (inference.py:176-180) whose four entries
    are lazy views of device-resident results.  Carries the session and the trial list so that
    learning.updateParams* can run the M-step without the data ever leaving HBM."""

    def __init__(self, session, trial_idx, local_pos):
        super().__init__()
        self.session = session
        self.trial_idx = np.asarray(trial_idx, dtype=np.int32)       # trials this rank processed
        self.stamp = session.post_stamp
        ctx = session.ctx
        n = len(self.trial_idx)
        tid = self.trial_idx
        self.local_positions = local_pos                              # positions in the caller's trial list
        self['post_mean'] = LazyTrialList(n, lambda i: self._fresh(i, 'post_mean') and ctx.post_mean(tid[i:i + 1])[0])
        self['post_vsm'] = LazyTrialList(n, lambda i: self._fresh(i, 'post_vsm') and ctx.post_vsm(tid[i:i + 1])[0])
        self['post_vsmGP'] = LazyTrialList(n, lambda i: self._fresh(i, 'post_vsmGP') and ctx.post_vsmgp(tid[i:i + 1])[0])
        self['post_cov'] = LazyTrialList(n, lambda i: self._fresh(i, 'post_cov') and ctx.post_cov(int(tid[i])))

    def _fresh(self, i, what):
        """The reference returns immutable per-trial arrays; here entry i is a view of device memory that the next
        E-step touching the same trial overwrites in place.  Reading it afterwards must fail, not return the newer
        posterior under the old name (entries already fetched, or snapshotted with materialize(), stay valid)."""
        if self.session.trial_stamp[self.trial_idx[i]] != self.stamp:
            raise _hip.HipBackendError("infRes['%s'][%d] belongs to a superseded E-step: trial %d has been overwritten on the "
                                       'device by a later one (call infRes.materialize() before running it to keep a host copy)'
                                       % (what, i, int(self.trial_idx[i])))
        return True

    def materialize(self, keys=('post_mean', 'post_vsm')):
        """Snapshot the listed entries to host arrays (bulk download) so that they survive later E-steps on the same
        session, like the reference's per-trial arrays do."""
        n = len(self.trial_idx)
        if n == 0:
            return self
        fetch = {'post_mean': self.session.ctx.post_mean, 'post_vsm': self.session.ctx.post_vsm, 'post_vsmGP': self.session.ctx.post_vsmgp}
        for key in keys:
            lst = self[key]
            missing = [i for i in range(n) if i not in lst._cache]
            if not missing:
                continue
            for i in missing:
                self._fresh(i, key)
            step = max(1, (1 << 28) // max(1, {'post_mean': self.session.p * self.session.T, 'post_vsm': self.session.T * self.session.p ** 2,
                                               'post_vsmGP': self.session.T ** 2 * self.session.p}[key] * 8))
            for c0 in range(0, len(missing), step):
                part = missing[c0:c0 + step]
                arr = fetch[key](self.trial_idx[part])
                for j, i in enumerate(part):
                    lst._cache[i] = arr[j].copy()
        return self

    def host_bytes(self, keys=('post_mean', 'post_vsm')):
        per = {'post_mean': self.session.p * self.session.T, 'post_vsm': self.session.T * self.session.p ** 2,
               'post_vsmGP': self.session.T ** 2 * self.session.p}
        return 8 * len(self.trial_idx) * sum(per[k] for k in keys)


class DeviceOptimRes(LazyTrialList):
    """lapOptimRes (inference.py:92,127): the modes, flattened; resident on device for warm starts."""

    def __init__(self, session, trial_idx):
        tid = np.asarray(trial_idx, dtype=np.int32)
        ctx = session.ctx
        super().__init__(len(tid), lambda i: self._fresh(i) and ctx.post_mean(tid[i:i + 1])[0].reshape(-1))
        self.session = session
        self.trial_idx = tid
        self.stamp = session.mode_stamp
        self.post_stamp = session.post_stamp

    def _fresh(self, i):
        if self.session.trial_stamp[self.trial_idx[i]] != self.post_stamp:
            raise _hip.HipBackendError('lapOptimRes[%d] belongs to a superseded E-step: the mode of trial %d has been overwritten '
                                       'on the device by a later one' % (i, int(self.trial_idx[i])))
        return True


class DeviceDualOptimRes(LazyTrialList):
    """varOptimRes of inference.dualVariational (inference.py:326 / 398: the optimal lambda, or rho = log lambda, per trial) left on the
    device: an entry is downloaded when it is read, and handed back as prevOptimRes of the next call on the same trials it is a warm start that
    moves no bytes (at config 5 a trial's entry is 4 MB: 1 GB per 256 trials each way).  Like the lazy infRes entries it is a VIEW: the next
    variational E-step on a trial overwrites its lambda in place, and reading the entry afterwards raises instead of returning the newer
    optimum under the old name (entries already read, or snapshotted with materialize(), stay valid)."""

    def __init__(self, session, trial_idx, log):
        tid = np.asarray(trial_idx, dtype=np.int32)
        ctx = session.ctx
        super().__init__(len(tid), lambda i: self._fresh(i) and (np.log(ctx.dual_lambda(tid[i:i + 1])[0]) if log else ctx.dual_lambda(tid[i:i + 1])[0]))
        self.session = session
        self.trial_idx = tid
        self.log = bool(log)
        self.stamp = session.mode_stamp
        self.dual_stamp = session.dual_stamp

    def _fresh(self, i):
        if self.session.dual_trial_stamp[self.trial_idx[i]] != self.dual_stamp:
            raise _hip.HipBackendError('varOptimRes[%d] belongs to a superseded variational E-step: the dual variables of trial %d have been '
                                       'overwritten on the device by a later one (call materialize() on the list before running it to keep a '
                                       'host copy)' % (i, int(self.trial_idx[i])))
        return True

    def materialize(self):
        """Host copies of every entry not read yet (one bulk download per 256 MB)."""
        missing = [i for i in range(len(self)) if i not in self._cache]
        if not missing:
            return self
        for i in missing:
            self._fresh(i)
        step = max(1, (1 << 28) // max(1, self.session.q * self.session.T * 8))
        for c0 in range(0, len(missing), step):
            part = missing[c0:c0 + step]
            lam = self.session.ctx.dual_lambda(self.trial_idx[part])
            for j, i in enumerate(part):
                self._cache[i] = np.log(lam[j]) if self.log else lam[j].copy()
        return self


# ----------------------------------------------------------------------------------------------
class Session:
    def __init__(self, Y, p, bin_ms):
        R, q, T = Y.shape
        self.R, self.q, self.T, self.p = R, q, T, p
        self.ctx = _hip.Context(q, p, T, R, bin_ms, device=WORLD.device())
        self.ctx.upload_counts(Y)
        self.post_stamp = 0
        self.mode_stamp = 0
        self.trial_stamp = np.zeros(R, dtype=np.int64)      # post_stamp of the E-step that last wrote each trial's posterior
        self.dual_stamp = 0
        self.dual_trial_stamp = np.zeros(R, dtype=np.int64)  # dual_stamp of the variational E-step that last wrote each trial's dual variables
        self.rank, self.size = 0, 1
        self.comm_ready = False
        if WORLD.enabled:
            import sys
            limit = float(os.environ.get('PGPFA_COMM_TIMEOUT', '300'))
            with _stdout_to_stderr():
                uid, path = WORLD.exchange_unique_id()
                # (every rank is known to be alive and to hold this id; what can still block is RCCL itself: bounded by a watchdog)
                with phase_deadline('pgpfa_comm_init (ncclCommInitRank)', limit, WORLD.rank):
                    self.ctx.comm_init(uid, WORLD.rank, WORLD.size)
                self.rank, self.size = WORLD.rank, WORLD.size
                self.comm_ready = True
                # first collective doubles as the barrier after which rank 0 may remove the file
                with phase_deadline('the first all-reduce', limit, WORLD.rank):
                    ones = self.ctx.allreduce_host(np.ones(1))
                # one line per rank for the launcher's log: which device, which PCI function, how many ranks RCCL itself saw
                sys.stderr.write('pgpfa: %s first_allreduce_sum %g\n' % (self.ctx.comm_describe(), float(ones[0])))
                sys.stderr.flush()
                if int(round(float(ones[0]))) != WORLD.size:
                    raise _hip.HipBackendError('first all-reduce over %d ranks summed to %g' % (WORLD.size, float(ones[0])))
            cleanup_rendezvous(path)

    def set_params(self, params):
        self.ctx.set_params(params['C'], params['d'], params['tau'])

    def mark_dual_written(self, trial_idx):
        """A variational E-step has just overwritten the resident dual variables of these trials."""
        self.dual_stamp += 1
        self.dual_trial_stamp[np.asarray(trial_idx, dtype=np.int64)] = self.dual_stamp

    def mark_written(self, trial_idx):
        """An E-step (or an uploaded posterior) has just overwritten the device state of these trials."""
        self.post_stamp += 1
        self.mode_stamp += 1
        self.trial_stamp[np.asarray(trial_idx, dtype=np.int64)] = self.post_stamp

    def local_slice(self, n_items):
        return shard_slice(n_items, self.rank, self.size)

    def allreduce(self, arr):
        return self.ctx.allreduce_host(arr) if self.comm_ready else np.asarray(arr, dtype=np.float64)


_sessions = weakref.WeakKeyDictionary()


def _stack_counts(experiment):
    Y = np.stack([np.asarray(tr['Y']) for tr in experiment.data])
    if Y.min() >= 0 and Y.max() <= 65535 and np.all(Y == np.floor(Y)):
        return Y.astype(np.uint8 if Y.max() <= 255 else np.uint16)
    return Y.astype(np.float64)            # (the C-ABI rejects it with the reason)


def session_for(experiment, p):
    """Return (session, trial indices of `experiment` inside the session's count tensor).

    Sub-sampled experiments produced by funs.util.subsampleTrials carry `_pgpfa_parent` and
    `batchTrIdx`, so minibatches reuse the parent's resident tensor instead of re-uploading."""
    parent = getattr(experiment, '_pgpfa_parent', None)
    if parent is not None and hasattr(experiment, 'batchTrIdx'):
        sess, _ = session_for(parent, p)
        return sess, np.asarray(experiment.batchTrIdx, dtype=np.int32)
    per_exp = _sessions.get(experiment)
    if per_exp is None:
        per_exp = {}
        _sessions[experiment] = per_exp
    n_trials = len(experiment.data)
    sess = per_exp.get(p)
    if sess is None or sess.R != n_trials:
        Y = _stack_counts(experiment)
        sess = Session(Y, p, float(experiment.binSize))
        per_exp[p] = sess
    return sess, np.arange(n_trials, dtype=np.int32)


def drop_sessions(experiment=None):
    """Free device memory held for `experiment` (or for every experiment)."""
    if experiment is None:
        for per_exp in list(_sessions.values()):
            for s in per_exp.values():
                s.ctx.close()
        _sessions.clear()
    elif experiment in _sessions:
        for s in _sessions[experiment].values():
            s.ctx.close()
        del _sessions[experiment]

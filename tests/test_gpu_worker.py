"""The worker's side of `bwa bam2bam -p` / `bwa worker` (reference bam2bam.c:1387-1442, 2099-2176) without a socket: nabwa_worker_*
take the master's messages and answer them through callbacks.  An in-process "master" sends the records of a mixed file (single reads
and pairs of two read groups) pristine, collects the positioned replies, builds the insert-size estimates from them as the master's
output thread does, broadcasts them, sends the positioned records again and must get back -- finished -- the very records the
two-pass front-end writes for the same file on one random stream.  Also: records sent twice, positioned records before any estimate,
finished records and end markers, the gathering loop with its timeouts, and the positioned state through the temporary-file route
(encode -> decode -> restore into a fresh batch)."""
import ctypes as C
import importlib
import os
import struct

import numpy as np
import pytest

import bamlib as B
import nabwa_testlib as T
import wirelib as W
from test_gpu_bam import bind as bind_bam, chk, toy_ann

nabwa = importlib.import_module("network-aware-bwa_amd")
pytestmark = pytest.mark.gpu
P = C.c_void_p
SEND = C.CFUNCTYPE(C.c_int, P, C.POINTER(C.c_uint8), C.c_int64)
RECV = C.CFUNCTYPE(C.c_int, P, C.POINTER(C.POINTER(C.c_uint8)), C.POINTER(C.c_int64), C.c_int)


def bind():
    L = W.bind(bind_bam())
    L.nabwa_worker_create.argtypes = [P, P, P, P]
    L.nabwa_worker_destroy.argtypes = [P]
    L.nabwa_worker_set_isize.argtypes = [P, P, C.c_int64]
    L.nabwa_worker_process.argtypes = [P, C.c_int, P, P, SEND, P]
    L.nabwa_worker_core.argtypes = [P, RECV, SEND, P, P]
    L.nabwa_worker_counts.argtypes = [P, P]
    L.nabwa_bam_batch_positioned.argtypes = [P, P]
    L.nabwa_bam_batch_restore.argtypes = [P, P]
    return L


@pytest.fixture(scope="module")
def world():
    L = bind()
    ix = nabwa.Index.load(T.TOY, 0, True, True)
    se = T.read_fastq(os.path.join(T.GOLDEN, "reads_se.fq"))[:200]
    pe = [T.read_fastq(os.path.join(T.GOLDEN, "reads_pe_%d.fq" % e)) for e in (1, 2)]
    logical = []                                           # (kind, [BAM records]) in file order: pairs of two read groups between single reads
    for i in range(300):
        if i % 3 == 0 and i // 3 < len(se):
            n, s, q = se[i // 3]
            logical.append((1, [B.make_record(n, s, q, 4)]))
        n, s1, q1 = pe[0][i]
        _, s2, q2 = pe[1][i]
        rg = B.tag_z("RG", "libA" if i % 2 else "libB")
        logical.append((2, [B.make_record(n, s1, q1, 1 | 64 | 4 | 8, rg), B.make_record(n, s2, q2, 1 | 128 | 4 | 8, rg)]))
    opt = nabwa.gap_init_opt()
    po = nabwa.pe_opt_default()
    yield dict(L=L, ix=ix, logical=logical, opt=opt, po=po)
    ix.close()


def direct(w):
    """the front-end's two passes over the whole file in one batch: the records and the table's blob"""
    L, ix = w["L"], w["ix"]
    l_pac, contigs = toy_ann()
    recs = [r for _, rs in w["logical"] for r in rs]
    buf, off = B.pack(recs)
    tab = P(L.nabwa_isize_table_create(w["po"].ap_prior, l_pac))
    st = C.c_uint64(nabwa.srand48_state(11))
    h = P()
    chk(L, L.nabwa_bam_batch_create(ix._h, C.byref(w["opt"]), C.byref(w["po"]), len(recs), T.ptr(buf), T.ptr(off), C.byref(h)))
    chk(L, L.nabwa_bam_batch_pass1(h, C.byref(st), tab))
    chk(L, L.nabwa_isize_table_infer_all(tab))
    n = L.nabwa_isize_table_encode(tab, None, 0)
    blob = np.zeros(n, np.uint8)
    assert L.nabwa_isize_table_encode(tab, T.ptr(blob), n) == n
    tot, mp = (C.c_uint64 * 2)(), (C.c_uint64 * 2)()
    chk(L, L.nabwa_bam_batch_pass2(h, tab, tot, mp))
    nb = C.c_int64()
    oo = np.zeros(len(recs) + 1, np.int64)
    L.nabwa_bam_batch_output(h, None, 0, T.ptr(oo), C.byref(nb))
    ob = np.zeros(nb.value, np.uint8)
    chk(L, L.nabwa_bam_batch_output(h, T.ptr(ob), nb.value, T.ptr(oo), C.byref(nb)))
    L.nabwa_bam_batch_destroy(h)
    L.nabwa_isize_table_destroy(tab)
    return [bytes(ob[oo[i]:oo[i + 1]]) for i in range(len(recs))], blob


class Transport:
    """what stands where the sockets would: a queue towards the worker, a list of what it sent back"""

    def __init__(self):
        self.inbox, self.sent, self.keep = [], [], None
        self.send = SEND(self._send)
        self.recv = RECV(self._recv)
        self.closed_after_empty = True

    def _send(self, ctx, p, n):
        self.sent.append(C.string_at(p, n))
        return 0

    def _recv(self, ctx, pm, pl, timeout_ms):
        if not self.inbox:
            return -1 if self.closed_after_empty else 0
        m = self.inbox.pop(0)
        self.keep = (C.c_uint8 * len(m)).from_buffer_copy(m)
        pm[0] = C.cast(self.keep, C.POINTER(C.c_uint8))
        pl[0] = len(m)
        return 1


def process(L, wk, msgs):
    t = Transport()
    arr = (P * len(msgs))()
    bufs = [(C.c_uint8 * len(m)).from_buffer_copy(m) for m in msgs]
    for i, b in enumerate(bufs):
        arr[i] = C.addressof(b)
    lens = (C.c_int64 * len(msgs))(*[len(m) for m in msgs])
    rc = L.nabwa_worker_process(wk, len(msgs), arr, lens, t.send, None)
    return rc, t.sent


def new_worker(w):
    wk = P()
    chk(w["L"], w["L"].nabwa_worker_create(w["ix"]._h, C.byref(w["opt"]), C.byref(w["po"]), C.byref(wk)))
    return wk


def pristine_messages(w):
    return [W.message(1000 + k, kind, 0, [dict(bam=r) for r in rs]) for k, (kind, rs) in enumerate(w["logical"])]


def bam_of(L, msg):
    """the reads of a message as they would stand in a BAM stream"""
    rc, rec, keep = W.decode(L, msg)
    assert rc == 0
    out = []
    for e in range(rec.kind):
        x = rec.read[e]
        core = (C.c_uint8 * 32)()
        L.nabwa_wire_core_to_bam(x.core, core)
        out.append(struct.pack("<I", 32 + x.data_len) + bytes(core) + C.string_at(x.data, x.data_len))
    return rec, out


def test_worker_answers_are_the_two_pass_front_end_records(world):
    L = world["L"]
    want, blob = direct(world)
    wk = new_worker(world)
    first = pristine_messages(world)
    # phase one, in three uneven batches: pair_aln + pair_posn
    positioned = []
    for lo, hi in ((0, 7), (7, 250), (250, len(first))):
        rc, sent = process(L, wk, first[lo:hi])
        assert rc == 0, L.nabwa_last_error()
        positioned += sent
    assert len(positioned) == len(first)
    for m0, m1 in zip(first, positioned):
        r0, _ = bam_of(L, m0)
        r1, _ = bam_of(L, m1)
        assert (r1.recno, r1.kind, r1.phase) == (r0.recno, r0.kind, 2)
    # positioned records before any estimate has arrived come back untouched and are counted
    rc, sent = process(L, wk, positioned[:5])
    assert rc == 0 and sent == positioned[:5]
    cnt = (C.c_uint64 * 4)()
    L.nabwa_worker_counts(wk, cnt)
    assert list(cnt) == [len(first), 0, 5, 0]
    # the master infers the insert sizes from the positioned records it has collected and broadcasts them; here: the blob of the direct run
    chk(L, L.nabwa_worker_set_isize(wk, T.ptr(blob), len(blob)))
    # phase two, other batch borders, one record sent twice, a finished record and an end marker in between
    again = positioned[:100] + [positioned[17]] + positioned[100:]
    rc, fin = process(L, wk, again[:60])
    assert rc == 0, L.nabwa_last_error()
    eof = W.message(7, 0, 0, [])
    rc, more = process(L, wk, again[60:] + [fin[3], eof])
    assert rc == 0, L.nabwa_last_error()
    assert more[-1] == eof and more[-2] == fin[3]                       # nothing to do: they go back as they came
    fin += more[:-2]
    assert fin[100] == fin[17]                                          # the resent record: the same answer again
    del fin[100]
    got = []
    for m in fin:
        rec, recs = bam_of(L, m)
        assert rec.phase == 3
        got += recs
    assert len(got) == len(want)
    for i, (g, x) in enumerate(zip(got, want)):
        assert g == x, i
    L.nabwa_worker_destroy(wk)


def test_worker_core_gathers_and_ends_on_a_quiet_transport(world):
    L = world["L"]
    wk = new_worker(world)
    t = Transport()
    t.inbox = pristine_messages(world)[:50]
    wo = (C.c_int32 * 3)(16, 0, 50)                                     # batches of at most 16, no waiting for more, idle after 50 ms
    t.closed_after_empty = False
    rc = L.nabwa_worker_core(wk, t.recv, t.send, None, wo)
    assert rc == 0, L.nabwa_last_error()
    assert len(t.sent) == 50
    for m0, m1 in zip(pristine_messages(world)[:50], t.sent):           # answered in arrival order
        assert bam_of(L, m0)[0].recno == bam_of(L, m1)[0].recno and bam_of(L, m1)[0].phase == 2
    # 1024 positioned records without an estimate end the worker (bam2bam.c:1428-1433)
    t2 = Transport()
    t2.inbox = [t.sent[0]] * 1100
    rc = L.nabwa_worker_core(wk, t2.recv, t2.send, None, wo)
    assert rc == nabwa.EIO and b"1024" in L.nabwa_last_error()
    L.nabwa_worker_destroy(wk)


def test_positioned_state_through_the_temporary_file_route(world):
    """pass 1 in one batch, its state written out as the reference's temporary file holds it (u32 length + message), read back into a
    fresh batch of the same records: pass 2 there gives the records of the batch that never left memory"""
    L, ix = world["L"], world["ix"]
    want, blob = direct(world)
    l_pac, _ = toy_ann()
    recs = [r for _, rs in world["logical"] for r in rs]
    kinds = [k for k, _ in world["logical"]]
    buf, off = B.pack(recs)
    tab = P(L.nabwa_isize_table_create(world["po"].ap_prior, l_pac))
    st = C.c_uint64(nabwa.srand48_state(11))
    h = P()
    chk(L, L.nabwa_bam_batch_create(ix._h, C.byref(world["opt"]), C.byref(world["po"]), len(recs), T.ptr(buf), T.ptr(off), C.byref(h)))
    chk(L, L.nabwa_bam_batch_pass1(h, C.byref(st), tab))
    state = (W.WireRead * len(recs))()
    chk(L, L.nabwa_bam_batch_positioned(h, state))
    nb = C.c_int64()
    oo = np.zeros(len(recs) + 1, np.int64)
    L.nabwa_bam_batch_output(h, None, 0, T.ptr(oo), C.byref(nb))
    ob = np.zeros(nb.value, np.uint8)
    chk(L, L.nabwa_bam_batch_output(h, T.ptr(ob), nb.value, T.ptr(oo), C.byref(nb)))
    spill = b""
    at = 0
    for k, kind in enumerate(kinds):
        rec = W.WireRec()
        rec.recno, rec.kind, rec.phase = k, kind, 2
        for e in range(kind):
            x = state[at]
            r = ob[oo[at]:oo[at + 1]]
            L.nabwa_wire_core_from_bam(T.ptr(r[4:36].copy()), x.core)
            x.data_len = len(r) - 36
            x.data = r[36:].ctypes.data
            rec.read[e] = x
            at += 1
        m = W.encode(L, rec)
        spill += struct.pack("<I", len(m)) + m
    L.nabwa_bam_batch_destroy(h)
    chk(L, L.nabwa_isize_table_infer_all(tab))
    # ... and back
    msgs, p = [], 0
    while p < len(spill):
        n = struct.unpack_from("<I", spill, p)[0]
        msgs.append(spill[p + 4:p + 4 + n])
        p += 4 + n
    assert len(msgs) == len(kinds)
    recs2, state2, keep = [], (W.WireRead * len(recs))(), []
    at = 0
    for m in msgs:
        rc, rec, buf_ = W.decode(L, m)
        assert rc == 0
        keep.append(buf_)
        _, rs = bam_of(L, m)
        recs2 += rs
        for e in range(rec.kind):
            state2[at] = rec.read[e]
            at += 1
    buf2, off2 = B.pack(recs2)
    h2 = P()
    chk(L, L.nabwa_bam_batch_create(ix._h, C.byref(world["opt"]), C.byref(world["po"]), len(recs2), T.ptr(buf2), T.ptr(off2), C.byref(h2)))
    chk(L, L.nabwa_bam_batch_restore(h2, state2))
    tot, mp = (C.c_uint64 * 2)(), (C.c_uint64 * 2)()
    chk(L, L.nabwa_bam_batch_pass2(h2, tab, tot, mp))
    L.nabwa_bam_batch_output(h2, None, 0, T.ptr(oo), C.byref(nb))
    ob2 = np.zeros(nb.value, np.uint8)
    chk(L, L.nabwa_bam_batch_output(h2, T.ptr(ob2), nb.value, T.ptr(oo), C.byref(nb)))
    got = [bytes(ob2[oo[i]:oo[i + 1]]) for i in range(len(recs2))]
    assert got == want
    # a batch made with other trimming options refuses the state: the lengths disagree
    o3 = nabwa.gap_init_opt()
    o3.trim_qual = 35
    h3 = P()
    chk(L, L.nabwa_bam_batch_create(ix._h, C.byref(o3), C.byref(world["po"]), len(recs2), T.ptr(buf2), T.ptr(off2), C.byref(h3)))
    rc = L.nabwa_bam_batch_restore(h3, state2)
    L.nabwa_bam_batch_destroy(h3)
    L.nabwa_bam_batch_destroy(h2)
    L.nabwa_isize_table_destroy(tab)
    assert rc in (0, nabwa.EINVAL)                     # (EINVAL wherever a read of the file is trimmed at q = 35)

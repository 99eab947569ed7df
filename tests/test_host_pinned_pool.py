"""Host side of the result arrays (ophelia_amd.engine.PinnedPool): recycling by size, release when the last view dies, the
fall-back to ordinary arrays when no pinned memory can be had (this container has no GPU: oph_host_alloc fails), and the
residency test the engine applies to arrays handed back to it.  No compute calls."""
import ctypes as C
import gc
import weakref

import numpy as np

from ophelia_amd import engine as E


class FakePool(E.PinnedPool):
    """PinnedPool over malloc'ed blocks instead of hipHostMalloc (the bookkeeping is what is under test)."""

    def __init__(self, keep=2):
        E.PinnedPool.__init__(self, keep)
        self.bufs, self.freed = {}, []

    def empty(self, shape, dtype=np.float32):
        dtype = np.dtype(dtype)
        nbytes = max(1, int(np.prod(shape)) * dtype.itemsize)
        lst = self.idle.get(nbytes)
        if lst:
            ptr = lst.pop()
        else:
            buf = (C.c_char * nbytes)()
            ptr = C.addressof(buf)
            self.bufs[ptr] = buf
        block = E._PinnedBlock(self, ptr, nbytes)
        return np.asarray(block)[:int(np.prod(shape)) * dtype.itemsize].view(dtype).reshape(shape)

    def _release(self, nbytes, ptr):
        lst = self.idle.setdefault(nbytes, [])
        if len(lst) < self.keep:
            lst.append(ptr)
        else:
            self.freed.append(ptr)


def test_blocks_return_to_the_pool_when_the_last_view_dies():
    pool = FakePool()
    a = pool.empty((4, 8))
    a[:] = 3.0
    v = a[1:3, ::2]                       # a view keeps the block alive
    ptr = a.ctypes.data
    del a
    gc.collect()
    assert pool.idle == {} and float(v.sum()) == 3.0 * 8
    del v
    gc.collect()
    assert pool.idle == {4 * 8 * 4: [ptr]}
    b = pool.empty((8, 4))                # same size: the block is reused
    assert b.ctypes.data == ptr and pool.idle[128] == []
    c = pool.empty((2, 2), np.int32)      # another size: its own block
    assert c.dtype == np.int32 and c.ctypes.data != ptr


def test_at_most_keep_idle_blocks_per_size_stay_allocated():
    pool = FakePool(keep=2)
    arrs = [pool.empty((16,)) for _ in range(4)]
    del arrs
    gc.collect()
    assert len(pool.idle[64]) == 2 and len(pool.freed) == 2


def test_without_pinned_memory_an_ordinary_array_is_returned():
    a = E.PINNED.empty((3, 5))            # no HIP device here: oph_host_alloc fails
    assert a.shape == (3, 5) and a.dtype == np.float32 and a.flags.writeable
    a[:] = 1.0


def test_residency_needs_the_very_same_read_only_array():
    eng = E.Engine.__new__(E.Engine)      # no handle: only the residency bookkeeping is exercised
    eng.resident_results = True
    K, V = np.zeros((2, 3)), np.ones((2, 3))
    eng._seal(K), eng._seal(V)
    token = (weakref.ref(K), weakref.ref(V))
    assert eng._is_resident(token, K, V)
    assert not eng._is_resident(token, K.copy(), V)          # a copy is not the resident array
    assert not eng._is_resident(token, V, K)                 # nor the two swapped
    assert not eng._is_resident(None, K, V)
    K.flags.writeable = True                                  # made writable again: may have been modified -> upload
    assert not eng._is_resident(token, K, V)
    eng.resident_results = False                              # residency switched off: nothing is sealed, nothing is resident
    W = np.zeros((2, 3))
    assert eng._seal(W).flags.writeable and not eng._is_resident((weakref.ref(W),), W)
    eng._h = None

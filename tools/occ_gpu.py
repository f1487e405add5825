import sys, os, numpy as np
sys.path.insert(0, '.')
from trio_binning_amd import kmers, _lib
import ctypes as C
lib=_lib.lib
n=40_000_000; k=21
def keys(seed):
    d=C.c_void_p(); _lib.check(lib.tbk_device_alloc(0, n*8, C.byref(d)))
    _lib.check(lib.tbk_synth_keys_device(0, seed, 0, n, k, d))
    h=np.empty(n,dtype=np.uint64); _lib.check(lib.tbk_memcpy_d2h(0, h.ctypes.data, d, n*8)); lib.tbk_device_free(0,d)
    return h
a=kmers.HashSet.from_keys(keys(11), k); b=kmers.HashSet.from_keys(keys(12), k)
c=kmers.Classifier(a,b)
s=c.stats()
print({x:s[x] for x in ('distinct_a','distinct_b','n_buckets','keys_behind_front','keys_past_half','front_layout','sampling_t','shared_keys')})
print('behind %', 100*s['keys_behind_front']/(s['distinct_a']+s['distinct_b']), 'lambda', s['distinct_a']/s['n_buckets'])

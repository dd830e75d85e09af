"""Debugging aid: which XCD the item workgroups of k_item_user land on, with and without the side stream beside them.
Needs a library built with -DHSK_DEBUG_XCC (see hsk_fused.hip: hsk_debug_xcc) at HSK_LIB_PATH.
usage: HSK_LIB_PATH=... python tools/xcc_hist.py [--no-prefetch]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from hassaku_amd import _lib

def main():
    prefetch = '--no-prefetch' not in sys.argv
    device = torch.device('cuda:0')
    lib = _lib.load()
    f = lib.hsk_debug_xcc
    f.argtypes = [ctypes.c_void_p, ctypes.c_int]; f.restype = ctypes.c_int
    out = np.zeros(64, dtype=np.uint32)
    r = bench.run_training('ml10m', device, 40, 34, prefetch=prefetch)
    torch.cuda.synchronize()
    assert f(out.ctypes.data, 1) == 0
    h = out.reshape(8, 8)
    print('prefetch' if prefetch else 'no prefetch', 'ms_per_step', round(r['ms_per_step'] * 1e3, 1))
    print('rows: workgroup index % 8, columns: XCC id')
    print(h)
    off = h.sum() - h.max(axis=1).sum()
    print('workgroups away from their label\'s majority XCD: %d of %d (%.2f %%)' % (off, h.sum(), 100.0 * off / max(1, h.sum())))

if __name__ == '__main__':
    main()

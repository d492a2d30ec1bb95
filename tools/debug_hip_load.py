import ctypes, os, sys
order = sys.argv[1]
def maps():
    seen = set()
    for l in open('/proc/self/maps'):
        p = l.split()[-1]
        if ('amdhip' in p or 'hsa-runtime' in p) and p not in seen:
            seen.add(p); print('   ', p)
def mine():
    lib = ctypes.CDLL(os.path.abspath('quadruped_control_amd/libqc_balance.so'))
    n = ctypes.c_int(-5)
    hip = ctypes.CDLL('libamdhip64.so.7')
    rc = hip.hipGetDeviceCount(ctypes.byref(n))
    print('  hipGetDeviceCount rc', rc, 'n', n.value)
    return lib
if order == 'torch_first':
    import torch; print('torch', torch.cuda.is_available(), torch.version.hip); maps()
    mine(); maps()
elif order == 'mine_first':
    mine(); maps()
    import torch; print('torch', torch.cuda.is_available()); maps()
else:
    mine(); maps()

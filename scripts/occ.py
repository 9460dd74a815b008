import ctypes, sys
sys.path.insert(0,'/root/repo')
from pindel_amd import binding
import torch
L = binding.lib()
a=ctypes.c_int(); b=ctypes.c_int(); l=ctypes.c_uint()
for small in (1,0):
    rc=L.pg_debug_occupancy(100, 6, small, ctypes.byref(a), ctypes.byref(b), ctypes.byref(l))
    print('small',small,'rc',rc,'close blocks/CU',a.value,'far',b.value,'lds',l.value)
p=torch.cuda.get_device_properties(0)
print(p.multi_processor_count, p.max_threads_per_multi_processor)

import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "lammps-spherharm_amd"))
import torch
from shpair import capi, shapes
import itertools
for (L, nq), split in itertools.product([(6,16),(4,10),(6,24),(6,32),(7,20),(7,24),(7,32),(8,16),(8,20),(8,24),(9,12),(9,16),(9,20),(9,24),(10,16),(10,20),(11,16),(12,12),(5,32),(4,32),(12,32),(10,24),(11,24),(9,32),(8,32),(11,20),(12,16)], [0, 1]):
    sp = capi.ShPair(0)
    sp.set_option("split", split)
    sp.settings(nq); sp.set_ntypes(1, 1); sp.set_shape(0, L, shapes.random_shape(L, 7)); sp.coeff("*", "*", 1000.0, 1.25)
    import numpy as np
    x = np.array([[0,0,0],[1.5,0,0]], dtype=np.float64); q = np.array([[1,0,0,0]]*2, dtype=np.float64)
    sp.set_neighbors_csr(np.array([0],dtype=np.int32), np.array([0,1],dtype=np.int32), np.array([1],dtype=np.int32))
    sp.compute(2, x, q, np.ones(2,dtype=np.int32), np.zeros(2,dtype=np.int32))
    k = sp.kernel_info()
    print(L, nq, "split option", split, "rows", k["ring_rows"], "lds", k["lds_bytes_per_wave"], "waves/cu", k["waves_per_cu"], "wpp", k["waves_per_pair"])
    sp.close()

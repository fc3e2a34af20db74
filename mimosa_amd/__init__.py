"""mimosa_amd — MI355X-native LiDAR geometric-factor hot path (deskew -> voxel-map k-NN ->
plane fit -> point-to-plane residual/Jacobian -> 6x6 Hessian) behind a C ABI.

Python here is tooling only (ctypes loader over libmimosa_hip.so, synthetic-world generator,
bench/test drivers).  The product is mimosa_amd/csrc (HIP kernels + C ABI) and mimosa_amd/host
(C++ mirror of the reference's LidarManager / Geometric / ICPFactor surface).
"""

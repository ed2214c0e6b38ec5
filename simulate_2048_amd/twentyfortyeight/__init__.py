"""
Drop-in mirror of the reference's ``twentyfortyeight`` package (NumPy 2048 engine + env class):
same module paths, function names, argument meaning, return types and error behaviour, with every
board computation done by the HIP kernels through the C-ABI (one launch per call -- this is the
compatibility surface; ``simulate_2048_amd.BatchedEnv2048`` is the throughput surface).
"""

/* oracle/refshim/cudashim: opencv2/core/cuda.hpp (GpuMat) is not needed by the kernels compiled here.  TEST INFRASTRUCTURE. */

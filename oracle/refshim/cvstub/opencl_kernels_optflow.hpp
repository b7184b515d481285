// the OpenCL kernel strings are not needed: HAVE_OPENCL is undefined

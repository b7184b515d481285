// no OpenCL in the host build of the reference CPU class

/* shadows the CUDA class header: surf.cpp does not use it */

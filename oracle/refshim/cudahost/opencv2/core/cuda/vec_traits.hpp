/* oracle/refshim/cudahost: unused by the host class.  TEST INFRASTRUCTURE. */

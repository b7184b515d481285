/* oracle/refshim/cudashim/cudashim.cpp -- TEST INFRASTRUCTURE.  Dynamic shared memory of the block being run (blocks run one
 * after the other, so one buffer serves them all). */
#include "cudashim.h"
#include <vector>
namespace cudashim {
static std::vector<unsigned long long> g_smem;
void *dynamic_smem() { return g_smem.data(); }
void set_dynamic_smem(size_t bytes) { g_smem.assign(bytes / 8 + 8, 0ull); }
}

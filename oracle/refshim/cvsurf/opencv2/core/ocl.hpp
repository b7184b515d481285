#include "../core.hpp"

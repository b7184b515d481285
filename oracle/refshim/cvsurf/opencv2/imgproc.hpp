#include "core.hpp"

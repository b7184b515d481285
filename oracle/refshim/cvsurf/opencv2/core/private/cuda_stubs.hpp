#include "../../core.hpp"

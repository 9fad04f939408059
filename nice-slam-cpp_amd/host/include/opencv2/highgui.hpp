// see opencv2/core.hpp (minimal stand-in)
#pragma once
#include "core.hpp"

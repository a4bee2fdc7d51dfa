// MOCK (see core.hpp)
#pragma once
#include "core.hpp"

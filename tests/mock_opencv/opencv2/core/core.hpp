// declaration-only mock, see ../../README.md
#pragma once
#include "../opencv.hpp"

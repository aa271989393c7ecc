#pragma once
#include "ATen.h"

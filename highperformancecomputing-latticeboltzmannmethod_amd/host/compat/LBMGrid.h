// Forwarding header: lets code written against the reference's "LBMGrid.h" compile against the MI355X backend.
#pragma once
#include "../lbm/grid.hpp"

// Forwarding header: lets code written against the reference's "LBMSolver.h" compile against the MI355X backend.
#pragma once
#include "../lbm/solver.hpp"

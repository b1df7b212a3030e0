// Forwarding header: lets code written against the reference's "LBMConfig.h" compile against the MI355X backend.
#pragma once
#include "../lbm/params.hpp"

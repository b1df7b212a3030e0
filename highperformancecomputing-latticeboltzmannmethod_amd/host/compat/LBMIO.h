// Forwarding header: lets code written against the reference's "LBMIO.h" compile against the MI355X backend.
#pragma once
#include "../lbm/io.hpp"

// Force-included (-include) in front of the reference's imglib.cpp when it is
// compiled for oracle/_ref: glibc's <cmath> defines the macro M_PI, and
// imglib.cpp:16 declares a constexpr variable of that name (the reference
// targets MSVC, whose <cmath> does not define it).  Nothing else is changed.
#include <cmath>
#include <stdexcept>
#undef M_PI

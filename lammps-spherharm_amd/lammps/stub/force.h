// COMPILE-CHECK SCAFFOLD (see lammps_stub.h) — stands in for LAMMPS' force.h in this image only.
#include "lammps_stub.h"

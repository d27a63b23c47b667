// COMPILE-CHECK SCAFFOLD (see lammps_stub.h) — stands in for LAMMPS' utils.h in this image only.
#include "lammps_stub.h"

// COMPILE-CHECK SCAFFOLD (see lammps_stub.h) — stands in for LAMMPS' update.h in this image only.
#include "lammps_stub.h"

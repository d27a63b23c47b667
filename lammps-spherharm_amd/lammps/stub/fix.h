// stub: see lammps_stub.h
#include "lammps_stub.h"

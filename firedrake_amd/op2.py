"""The PyOP2-compatible API surface of the MI355X backend (mirror of pyop2/op2.py:42-69)."""
from .configuration import configuration  # noqa: F401
from .op2types import (Set, ExtrudedSet, Subset, DataSet, Dat, Global, Constant, Map, PermutedMap, ComposedMap,  # noqa: F401
                       Sparsity, Mat, Access, IterationRegion,
                       READ, WRITE, RW, INC, MIN, MAX, ON_BOTTOM, ON_TOP, ON_INTERIOR_FACETS, ALL,
                       IntType, ScalarType, MapValueError, ModeValueError, DataValueError, DataTypeError,
                       SetTypeError, SizeTypeError, SubsetIndexOutOfBounds)
from .kernel import (Kernel, CStringLocalKernel, GlobalKernel, GlobalKernelArg, DatKernelArg,  # noqa: F401
                     MatKernelArg, MapKernelArg, PermutedMapKernelArg)
from .parloop import (Parloop, ParLoop, LegacyParloop, parloop, par_loop, DatParloopArg,  # noqa: F401
                      GlobalParloopArg, MatParloopArg, DatLegacyArg, GlobalLegacyArg, MatLegacyArg)
from .compilation import CompilationError  # noqa: F401
from ._lib import FDHipError  # noqa: F401

"""The PyOP2-compatible API surface of the MI355X backend (mirror of pyop2/op2.py:42-69)."""
from .configuration import configuration  # noqa: F401
from .op2types import (Set, ExtrudedSet, Subset, MixedSet, DataSet, MixedDataSet, Dat, DatView, MixedDat, Global, Constant,  # noqa: F401
                       Map, PermutedMap, ComposedMap, MixedMap,
                       Sparsity, Mat, Access, IterationRegion,
                       READ, WRITE, RW, INC, MIN, MAX, ON_BOTTOM, ON_TOP, ON_INTERIOR_FACETS, ALL,
                       IntType, ScalarType, MapValueError, ModeValueError, DataValueError, DataTypeError,
                       SetTypeError, SizeTypeError, SubsetIndexOutOfBounds)
from .kernel import (Kernel, CStringLocalKernel, LoopyLocalKernel, loopy_c_kernel, GlobalKernel, GlobalKernelArg, DatKernelArg,  # noqa: F401
                     MatKernelArg, MapKernelArg, PermutedMapKernelArg, MixedDatKernelArg, MixedMatKernelArg,
                     PassthroughKernelArg)
from .parloop import (Parloop, ParLoop, LegacyParloop, parloop, par_loop, DatParloopArg,  # noqa: F401
                      GlobalParloopArg, MatParloopArg, MixedDatParloopArg, MixedMatParloopArg, DatLegacyArg,
                      GlobalLegacyArg, MatLegacyArg, MixedDatLegacyArg, MixedMatLegacyArg)
from .compilation import CompilationError  # noqa: F401
from ._lib import FDHipError  # noqa: F401

"""Exception types of the PyOP2-shaped API: the names and base classes of pyop2/exceptions.py:36-158 (user code and
the reference's tests catch these), plus the backend's own FDHipError (firedrake_amd/_lib.py) and CompilationError
(firedrake_amd/compilation.py)."""


class DataTypeError(TypeError):
    """Invalid type for data."""


class DimTypeError(TypeError):
    """Invalid type for dimension."""


class ArityTypeError(TypeError):
    """Invalid type for arity."""


class IndexTypeError(TypeError):
    """Invalid type for index."""


class NameTypeError(TypeError):
    """Invalid type for name."""


class SetTypeError(TypeError):
    """Invalid type for a Set."""


class SizeTypeError(TypeError):
    """Invalid type for size."""


class SubsetIndexOutOfBounds(TypeError):
    """Out of bound index."""


class SparsityTypeError(TypeError):
    """Invalid type for a Sparsity."""


class MapTypeError(TypeError):
    """Invalid type for a Map."""


class DataSetTypeError(TypeError):
    """Invalid type for a DataSet."""


class MatTypeError(TypeError):
    """Invalid type for a Mat."""


class DatTypeError(TypeError):
    """Invalid type for a Dat."""


class KernelTypeError(TypeError):
    """Invalid type for a Kernel."""


class DataValueError(ValueError):
    """Illegal value for data."""


class IndexValueError(ValueError):
    """Illegal value for index."""


class ModeValueError(ValueError):
    """Illegal value for mode."""


class IterateValueError(ValueError):
    """Illegal value for iterate."""


class SetValueError(ValueError):
    """Illegal value for a Set."""


class MapValueError(ValueError):
    """Illegal value for a Map."""


class ConfigurationError(RuntimeError):
    """Illegal configuration value or type."""


class SparsityFormatError(ValueError):
    """Unable to produce a sparsity for this matrix format."""


class CachingError(ValueError):
    """A caching error."""


class HashError(CachingError):
    """Something is wrong with the hash."""

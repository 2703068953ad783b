"""Exception types of the PyOP2-shaped API.

The class NAMES and their base classes are interface: user code and the reference's tests catch them by name
(pyop2/exceptions.py:36-158).  They are kept as one table and the classes are created from it; nothing here is logic.
The backend's own FDHipError lives in firedrake_amd/_lib.py, CompilationError in firedrake_amd/compilation.py."""

# (name, base, what the argument check that raises it found)
_INTERFACE = (
    ("DataTypeError", TypeError, "data of a type the carrier cannot hold"),
    ("DimTypeError", TypeError, "a dimension that is not an integer or a tuple of integers"),
    ("ArityTypeError", TypeError, "a Map arity that is not an integer"),
    ("IndexTypeError", TypeError, "an index of the wrong type"),
    ("NameTypeError", TypeError, "a name that is not a string"),
    ("SetTypeError", TypeError, "something other than a Set where a Set is required"),
    ("SizeTypeError", TypeError, "a size that is not an integer (or a tuple of them)"),
    ("SubsetIndexOutOfBounds", TypeError, "a Subset index outside its parent set"),
    ("SparsityTypeError", TypeError, "something other than a Sparsity where one is required"),
    ("MapTypeError", TypeError, "something other than a Map where one is required"),
    ("DataSetTypeError", TypeError, "something other than a DataSet where one is required"),
    ("MatTypeError", TypeError, "something other than a Mat where one is required"),
    ("DatTypeError", TypeError, "something other than a Dat where one is required"),
    ("KernelTypeError", TypeError, "something other than a Kernel where one is required"),
    ("DataValueError", ValueError, "data whose shape or values do not fit the carrier"),
    ("IndexValueError", ValueError, "an index outside the valid range"),
    ("ModeValueError", ValueError, "an access mode the operation does not accept"),
    ("IterateValueError", ValueError, "an iteration region that does not exist"),
    ("SetValueError", ValueError, "a Set that does not match (e.g. a Map's source against the iteration set)"),
    ("MapValueError", ValueError, "a Map that does not match the data it is used with"),
    ("ConfigurationError", RuntimeError, "an unknown configuration key or a value of the wrong type"),
    ("SparsityFormatError", ValueError, "a matrix format no sparsity can be built for"),
    ("CachingError", ValueError, "an object cache used inconsistently"),
    ("HashError", "CachingError", "a cache key that cannot be computed"),
)

__all__ = [row[0] for row in _INTERFACE]
for _name, _base, _what in _INTERFACE:
    _base = globals()[_base] if isinstance(_base, str) else _base
    globals()[_name] = type(_name, (_base,), {"__doc__": f"Raised for {_what}.", "__module__": __name__})
del _name, _base, _what

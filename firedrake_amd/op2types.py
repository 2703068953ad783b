"""PyOP2-shaped data carriers with device (HBM) mirrors.

Host-side mirror of pyop2/types/*.py, restricted to what the assembly hot path uses
(SURVEY.md 2.1): Set/ExtrudedSet/Subset (set.py:18-543), DataSet (dataset.py:17-199),
Dat (dat.py:27-711), Global (glob.py:21-480), Map/PermutedMap (map.py:17-470),
Sparsity/Mat (mat.py:27-985).  Same names, argument meaning and error behaviour, so the
tests in tests/ read like the reference's tests/pyop2/*.py.

Every carrier keeps a numpy host copy (what users see through ``.data``) and a device
mirror that the wrapper kernels read/write; validity flags decide when to copy
(the reference's ``dat_version``/``halo_valid`` bookkeeping, dat.py:154-173, :622-678,
extended by a host/device pair).
"""
from __future__ import annotations

import ctypes
import enum
from contextlib import contextmanager
from typing import Optional, Sequence

import numpy as np

from . import _lib
from .configuration import configuration
from .device import DeviceBuffer

IntType = np.dtype(np.int32)       # pyop2/datatypes.py:6-8 (32-bit PETSc indices)
ScalarType = np.dtype(np.float64)  # tsfc/parameters.py:19
RealType = ScalarType


class Access(enum.IntEnum):        # pyop2/types/access.py:4-37
    READ = 1
    WRITE = 2
    RW = 3
    INC = 4
    MIN = 5
    MAX = 6


READ, WRITE, RW, INC, MIN, MAX = (Access.READ, Access.WRITE, Access.RW, Access.INC, Access.MIN, Access.MAX)


class IterationRegion(enum.IntEnum):
    ON_BOTTOM = 1
    ON_TOP = 2
    ON_INTERIOR_FACETS = 3
    ALL = 4


ON_BOTTOM, ON_TOP, ON_INTERIOR_FACETS, ALL = (IterationRegion.ON_BOTTOM, IterationRegion.ON_TOP,
                                              IterationRegion.ON_INTERIOR_FACETS, IterationRegion.ALL)


# ---- exceptions (pyop2/exceptions.py) -------------------------------------------------
from .exceptions import (ArityTypeError, DataSetTypeError, DataTypeError, DataValueError, DatTypeError, DimTypeError,  # noqa: E402,F401
                         IndexTypeError, IndexValueError, MapTypeError, MapValueError, MatTypeError, ModeValueError,
                         NameTypeError, SetTypeError, SizeTypeError, SparsityTypeError, SubsetIndexOutOfBounds)


def _check_name(name):
    if name is not None and not isinstance(name, str):
        raise NameTypeError(f"name must be a string, got {type(name)}")


# ---- sets -----------------------------------------------------------------------------
class Set:
    """pyop2/types/set.py:18-121.  ``size`` is an int or the cumulative triple
    ``(core, owned, total)`` (set.py:32-55): entities [0,core) touch no ghost data,
    [core,owned) are owned but adjacent to ghosts, [owned,total) are ghosts that are
    never executed (set.py:115-121)."""

    _extruded = False
    _extruded_periodic = False
    _kernel_args_ = ()

    def __init__(self, size, name=None, halo=None, comm=None):
        _check_name(name)
        if isinstance(size, (int, np.integer)):
            size = [int(size)] * 3
        try:
            size = [int(s) for s in size]
        except (TypeError, ValueError):
            raise SizeTypeError(f"Set size must be an int or a (core, owned, total) triple, got {size!r}")
        if isinstance(size, str) or len(size) != 3 or not (0 <= size[0] <= size[1] <= size[2]):
            raise SizeTypeError(f"Set size must be an int or (core, owned, total) with core<=owned<=total, got {size}")
        self._sizes = tuple(size)
        self.name = name or f"set_#x{id(self):x}"
        self.halo = halo
        self.comm = comm

    @property
    def core_size(self):
        return self._sizes[0]

    @property
    def size(self):
        return self._sizes[1]

    @property
    def total_size(self):
        return self._sizes[2]

    @property
    def sizes(self):
        return self._sizes

    @property
    def core_part(self):          # set.py:115-117: (offset, size)
        return (0, self.core_size)

    @property
    def owned_part(self):         # set.py:119-121
        return (self.core_size, self.size - self.core_size)

    @property
    def superset(self):
        return self

    def __pow__(self, e):
        return DataSet(self, dim=e)

    def __len__(self):
        return 1

    def __iter__(self):
        yield self

    def __getitem__(self, idx):           # set.py:155-159
        if idx != 0:
            raise IndexTypeError("Can only extract component 0 from %r" % self)
        return self

    # set.py:143-149: Sets compare by sizes and name (default names are unique per object)
    def __hash__(self):
        return hash((type(self).__name__, self._sizes, self.name))

    def __eq__(self, other):
        return isinstance(other, Set) and not isinstance(other, (ExtrudedSet, Subset)) and type(self) is type(other) \
            and self._sizes == other._sizes and self.name == other.name

    def __ne__(self, other):
        return not self == other

    def __contains__(self, dset):         # set.py:183-189
        return isinstance(dset, DataSet) and dset.set is self

    def __call__(self, *indices):         # set.py:170-181: set(i, j, ...) -> Subset
        if len(indices) == 1:
            indices = indices[0]
            if np.isscalar(indices):
                indices = [indices]
        return Subset(self, indices)

    layers = property(lambda self: 1)     # set.py:197-199

    # -- set algebra between a Set and itself / its Subsets: see _set_algebra below
    def intersection(self, other):
        return _set_algebra(self, other, "and")

    def union(self, other):
        return _set_algebra(self, other, "or")

    def difference(self, other):
        return _set_algebra(self, other, "andnot")

    def symmetric_difference(self, other):
        return _set_algebra(self, other, "xor")

    def __str__(self):
        return "OP2 Set: %s with size %s" % (self.name, self.size)

    def __repr__(self):
        return f"Set({self._sizes!r}, {self.name!r})"


class ExtrudedSet(Set):
    """Constant-layer extruded set (pyop2/types/set.py:307-393).  ``layers`` counts node
    levels = cell layers + 1 (set.py:320-345); the kernel argument is [[0, layers]]
    (set.py:342-345, 351-353)."""

    _extruded = True

    def __init__(self, parent, layers, extruded_periodic=False):
        if not isinstance(parent, Set):
            raise TypeError("ExtrudedSet needs a parent Set")          # set.py:325 (validate_type)
        if isinstance(layers, (int, np.integer)):
            if layers < 2:
                raise SizeTypeError("Number of layers must be > 1 (not %s)." % layers)
            self._layers_array = np.array([[0, int(layers)]], dtype=IntType)
            self.constant_layers = True
        else:
            # set.py:326-337: one [bottom, top) row of node levels per entity of the parent set
            try:
                arr = np.ascontiguousarray(np.asarray(layers, dtype=IntType).reshape(parent.total_size, 2))
            except (TypeError, ValueError):
                raise SizeTypeError(f"Specifying layers per entity, but provided {np.shape(layers)}, "
                                    f"needed ({parent.total_size}, 2)")
            if arr.size and arr.min() < 0:
                raise SizeTypeError("Bottom of layers must be >= 0")
            if (arr[:, 1] - arr[:, 0] < 1).any():
                raise SizeTypeError("Number of layers must be >= 0")
            self._layers_array = arr
            self.constant_layers = False
        self._parent = parent
        self._sizes = parent._sizes
        self.name = parent.name + "_extruded"
        self.halo = parent.halo
        self.comm = parent.comm
        self._extruded_periodic = extruded_periodic
        self._dev_layers = None

    @property
    def parent(self):
        return self._parent

    def __contains__(self, set_):         # set.py:368-369
        return set_ is self._parent

    def __eq__(self, other):
        return self is other

    __hash__ = object.__hash__

    def __str__(self):
        return "OP2 ExtrudedSet: %s with size %s (%s layers)" % (self.name, self.size, self._layers_array)

    def __repr__(self):
        return "ExtrudedSet(%r, %r)" % (self._parent, self.layers if self.constant_layers else self._layers_array)

    @property
    def layers(self):                      # set.py:383-389
        if not self.constant_layers:
            raise ValueError("No single layer, use layers_array attribute")
        return int(self._layers_array[0, 1])

    @property
    def layers_array(self):
        return self._layers_array

    def _layers_dev(self):
        if self._dev_layers is None:
            self._dev_layers = DeviceBuffer.from_numpy(self._layers_array)
        return self._dev_layers.ptr


class Subset(Set):
    """pyop2/types/set.py:396-543: iterate only ``indices`` of ``superset``."""

    def __init__(self, superset, indices):
        if not isinstance(superset, Set):
            raise TypeError("Subset needs a Set")                      # set.py:407 (validate_type)
        if isinstance(superset, Subset):
            indices = superset.indices[np.asarray(indices, dtype=IntType)]
            superset = superset.superset
        self._superset = superset
        idx = np.unique(np.asarray(indices, dtype=IntType).reshape(-1))
        if len(idx) and (idx[0] < 0 or idx[-1] >= superset.total_size):
            raise SubsetIndexOutOfBounds("Out of bounds indices in Subset construction: [%d, %d) not [0, %d)" %
                                         (idx[0], idx[-1], superset.total_size))
        self._indices = idx
        self._sizes = ((idx < superset.core_size).sum(), (idx < superset.size).sum(), len(idx))
        self._sizes = tuple(int(s) for s in self._sizes)
        self.name = superset.name + "_subset"
        self.halo = superset.halo
        self.comm = superset.comm
        self._extruded = superset._extruded
        self._extruded_periodic = superset._extruded_periodic
        self.constant_layers = getattr(superset, "constant_layers", True)
        self._dev_indices = None

    @property
    def superset(self):
        return self._superset

    @property
    def indices(self):
        return self._indices

    @property
    def owned_indices(self):              # set.py:485-490
        return self._indices[self._indices < self._superset.size]

    def __eq__(self, other):
        return self is other

    __hash__ = object.__hash__

    intersection = Set.intersection      # membership-mask algebra shared with Set (_set_algebra)
    union = Set.union
    difference = Set.difference
    symmetric_difference = Set.symmetric_difference

    def __call__(self, *indices):         # set.py:462-473: a Subset of a Subset
        if len(indices) == 1:
            indices = indices[0]
            if np.isscalar(indices):
                indices = [indices]
        return Subset(self, indices)

    def __str__(self):
        return "OP2 Subset: %s with sizes %s" % (self.name, self._sizes)

    def __repr__(self):
        return "Subset(%r, %r)" % (self._superset, self._indices)

    @property
    def layers_array(self):
        return self._superset.layers_array

    @property
    def layers(self):
        return self._superset.layers

    def _layers_dev(self):
        return self._superset._layers_dev()

    def _indices_dev(self):
        if self._dev_indices is None:
            self._dev_indices = DeviceBuffer.from_numpy(self._indices)
        return self._dev_indices.ptr


def _set_algebra(lhs, rhs, op):
    """Set operations among a Set and its Subsets (the semantics of pyop2/types/set.py:201-232, 498-543), computed on
    membership masks over the common superset.  A result that covers the whole superset is the superset itself, anything
    else a Subset of it; operands that do not share a superset are a TypeError (two unrelated plain Sets: ValueError)."""
    for x in (lhs, rhs):
        if type(x) not in (Set, Subset):
            raise TypeError(f"set operations are defined between a Set and its Subsets, not {type(x)}")
    root = lhs.superset
    if rhs.superset is not root:
        if type(lhs) is Set and type(rhs) is Set:
            raise ValueError(f"{lhs} and {rhs} are unrelated sets")
        raise TypeError(f"{lhs} and {rhs} do not share a superset")

    def mask(x):
        m = np.zeros(root.total_size, dtype=bool)
        m[x._indices if type(x) is Subset else slice(None)] = True
        return m

    a, b = mask(lhs), mask(rhs)
    out = {"and": a & b, "or": a | b, "andnot": a & ~b, "xor": a ^ b}[op]
    if out.all() and root.total_size > 0:
        return root
    for x in (lhs, rhs):                         # hand an operand back unchanged when it already is the answer
        if type(x) is Subset and np.array_equal(out, mask(x)):
            return x
    return Subset(root, np.nonzero(out)[0].astype(IntType))


class DataSet:
    """pyop2/types/dataset.py:17-112: Set x dim."""

    def __init__(self, iter_set, dim=1, name=None, apply_local_global_filter=False):
        _check_name(name)
        if isinstance(iter_set, DataSet):
            dim = iter_set.dim
            iter_set = iter_set.set
        if isinstance(iter_set, Subset):
            raise NotImplementedError("Deriving a DataSet from a Subset is unsupported")
        if not isinstance(iter_set, Set):
            raise SetTypeError(f"expected a Set, got {type(iter_set)}")
        if isinstance(dim, (int, np.integer)):
            dim = (int(dim),)
        try:                                                       # dataset.py:27 (validate_type DimTypeError)
            if isinstance(dim, str) or not all(isinstance(d, (int, np.integer)) for d in dim):
                raise TypeError
            self._dim = tuple(int(d) for d in dim)
        except TypeError:
            raise DimTypeError(f"dim must be an int or a tuple of ints, got {dim!r}")
        self._set = iter_set
        self._cdim = int(np.prod(self._dim))
        self.name = name or f"dset_#x{id(self):x}"
        self._apply_local_global_filter = apply_local_global_filter

    @property
    def set(self):
        return self._set

    @property
    def dim(self):
        return self._dim

    @property
    def cdim(self):
        return self._cdim

    @property
    def size(self):
        return self._set.size

    @property
    def total_size(self):
        return self._set.total_size

    def __iter__(self):            # dataset.py:95-103: a DataSet is a one-member bag of itself
        yield self

    def __len__(self):
        return 1

    def __getitem__(self, idx):
        if idx != 0:
            raise IndexError("Can only extract component 0 from a DataSet")
        return self

    def __eq__(self, o):
        return isinstance(o, DataSet) and o._set is self._set and o._dim == self._dim

    def __ne__(self, o):
        return not self == o

    def __hash__(self):
        return hash((id(self._set), self._dim))

    def __contains__(self, dat):           # dataset.py:109-111
        return getattr(dat, "dataset", None) == self

    def __str__(self):
        return "OP2 DataSet: %s on set %s, with dim %s, %s" % (self.name, self._set, self._dim, self._apply_local_global_filter)

    def __repr__(self):
        return "DataSet(%r, %r, %r, %r)" % (self._set, self._dim, self.name, self._apply_local_global_filter)


def _as_dataset(x, dim=1):
    if isinstance(x, DataSet):
        return x
    if isinstance(x, Set):
        return DataSet(x, dim)
    raise DataSetTypeError(f"expected Set or DataSet, got {type(x)}")


class MixedDataSet:
    """pyop2/types/dataset.py:295-450: a bag of DataSets, built from a MixedSet (with dims) or from an iterable of
    Sets / DataSets."""

    def __init__(self, arg, dims=None):
        if isinstance(arg, MixedDataSet):
            dsets = arg.split
        elif isinstance(arg, str):
            raise DataSetTypeError("MixedDataSet needs a MixedSet or an iterable of Sets / DataSets")
        elif dims is not None:
            # dataset.py:364-374: a MixedSet / iterable of Sets with a scalar dim or one dim per Set
            sets = arg.split if isinstance(arg, MixedSet) else tuple(arg)
            if not all(isinstance(s_, Set) for s_ in sets):
                raise TypeError("with dims given, the first argument must be a MixedSet or an iterable of Sets")
            dims = (dims,) * len(sets) if isinstance(dims, (int, np.integer)) else tuple(dims)
            if len(sets) != len(dims):
                raise ValueError("Got MixedSet of %d Sets but %s dims" % (len(sets), len(dims)))
            dsets = tuple(s_ ** d for s_, d in zip(sets, dims))
        else:
            dsets = tuple(x if isinstance(x, DataSet) else _as_dataset(x) for x in arg)
        if not dsets:
            raise DataSetTypeError("MixedDataSet needs at least one DataSet")
        self._dsets = dsets
        self.comm = dsets[0].set.comm

    split = property(lambda self: self._dsets)
    dim = property(lambda self: tuple(d.dim for d in self._dsets))
    cdim = property(lambda self: sum(d.cdim for d in self._dsets))
    name = property(lambda self: tuple(d.name for d in self._dsets))          # dataset.py:409-412

    @property
    def set(self):
        return MixedSet(d.set for d in self._dsets)

    def __getitem__(self, idx):
        return self._dsets[idx]

    def __iter__(self):
        return iter(self._dsets)

    def __len__(self):
        return len(self._dsets)

    def __eq__(self, o):
        return isinstance(o, MixedDataSet) and self._dsets == o._dsets

    def __ne__(self, o):
        return not self == o

    def __hash__(self):
        return hash(self._dsets)

    def __str__(self):
        return "OP2 MixedDataSet composed of DataSets: %s" % (self._dsets,)

    def __repr__(self):
        return "MixedDataSet(%r)" % (self._dsets,)


# ---- host/device mirrored array ---------------------------------------------------------
# graph.CapturedStep sets this to (read, written) dictionaries {id: carrier} while it records a step: the bookkeeping below runs at
# capture time only, so the replay has to redo what it would have done -- bring the device copy of a carrier the step READS up to
# date when the host wrote it in between, and declare the host copy of every carrier the step WRITES stale
_capture_log = None


class _Mirrored:
    """numpy host array + device mirror with validity flags."""

    def _init_storage(self, host: np.ndarray):
        # the ONE buffer every handed-out view aliases -- always privately owned: were it the caller's array, the caller's
        # reference would look like a live view for ever (``_views_alive``) and one ``dat.data`` access would switch the carrier
        # to eager mirroring for good
        self._host = np.array(host, order="C", copy=True)
        self._dev: Optional[DeviceBuffer] = None
        self._host_valid = True
        self._dev_valid = False
        self._rw_handed = False
        self.dat_version = 0

    # Host/device coherence.  The reference has ONE buffer: every array ``dat.data`` ever returned aliases it, and code
    # that holds such a view across parloops keeps working (pyop2/types/dat.py:145-204).  Here the host buffer
    # ``_host`` is allocated once and never rebound -- downloads go INTO it, so views stay attached -- and the views
    # handed out are plain ndarray views whose ``base`` collapses to ``_host``: while any of them is alive the
    # reference count of ``_host`` says so.  Once a WRITABLE view has been handed out and views are still alive, the
    # host may be written at any moment without this class hearing of it; the Dat is then kept coherent the expensive
    # way: uploaded again before every device use, downloaded again after every device write (Parloop calls
    # ``_after_device_write``).  When the last view dies the Dat returns to lazy mirroring.  Read-only views (``data_ro``) are
    # refreshed the same way only while a writable one is alive; a read-only view kept across a device write shows the
    # values of the moment it was taken until ``data_ro`` is read again.
    def _views_alive(self) -> bool:
        import sys
        return sys.getrefcount(self._host) > 2               # our attribute + getrefcount's argument

    def _to_host(self):
        if not self._host_valid:
            from .device import wait_pending_graph
            wait_pending_graph()           # (a replayed graph writes on its own stream)
            if self._host.nbytes:
                _lib.call("fd_memcpy_d2h", self._host.ctypes.data, self._dev.ptr, self._host.nbytes, None)
            self._host_valid = True
        return self._host

    def _dev_ptr(self, write: bool) -> int:
        """Device pointer for a kernel; uploads if the host copy is newer (or may be: see above)."""
        if _capture_log is not None:                          # (graph.CapturedStep: the carriers a recorded step reads and writes)
            _capture_log[1 if write else 0][id(self)] = self
        if self._dev is None:
            self._dev = DeviceBuffer(self._host.nbytes)
            self._dev_valid = False
        if self._rw_handed and self._host_valid and self._dev_valid:
            if self._views_alive():
                self._dev_valid = False                       # a live writable view: assume it was written
                self.dat_version += 1                         # ... so nothing cached on the old values may be reused
            else:
                self._rw_handed = False
        if not self._dev_valid:
            self._dev.upload(self._to_host())
            self._dev_valid = True
        if write:
            self._host_valid = False
            self.dat_version += 1
        return self._dev.ptr

    def _after_device_write(self):
        """Called once the kernels that wrote this carrier are queued: keep live writable views current."""
        if self._rw_handed and not self._host_valid:
            if self._views_alive():
                self._to_host()
            else:
                self._rw_handed = False

    def _host_rw(self):
        h = self._to_host()
        self._dev_valid = False
        self._rw_handed = True
        self.dat_version += 1
        return h


class Dat(_Mirrored):
    """pyop2/types/dat.py:27-711.  Shape (total_size, *dim), ghosts at the tail (dat.py:83)."""

    def __init__(self, dataset, data=None, dtype=None, name=None):
        _check_name(name)
        if isinstance(dataset, Dat):
            data = dataset.data_ro_with_halos.copy() if data is None else data
            dtype = dataset.dtype if dtype is None else dtype
            dataset = dataset.dataset
        dataset = _as_dataset(dataset)
        self._dataset = dataset
        shape = (dataset.total_size,) + (() if dataset.dim == (1,) else dataset.dim)
        if dtype is not None:
            try:
                dtype = np.dtype(dtype)
            except TypeError:
                raise DataTypeError("Invalid data type: %s" % (dtype,))          # utils.verify_reshape
        if data is None:
            dt = np.dtype(dtype) if dtype is not None else ScalarType
            host = np.zeros(shape, dtype=dt)
        else:
            a = np.asarray(data, dtype=dtype)
            dt = a.dtype if dtype is None else np.dtype(dtype)
            try:
                host = np.array(a, dtype=dt).reshape(shape)
            except ValueError:
                raise DataValueError("Invalid data: expected %d values, got %d!" % (int(np.prod(shape)), a.size))
        self._init_storage(host)
        self.name = name or f"dat_#x{id(self):x}"
        self.halo_valid = True
        self._halo_frozen = False
        self._frozen_access_mode = None

    # -- container protocol (dat.py:113-122, 333-347): a Dat is a one-member bag of itself
    def __getitem__(self, idx):
        if idx != 0:
            raise IndexValueError("Can only extract component 0 from %r" % self)
        return self

    @property
    def split(self):
        return (self,)

    def __iter__(self):
        yield self

    def __len__(self):
        return 1

    def __str__(self):
        return "OP2 Dat: %s on (%s) with datatype %s" % (self.name, self._dataset, self.dtype.name)

    def __repr__(self):
        return "Dat(%r, None, %r, %r)" % (self._dataset, self.dtype, self.name)

    @property
    def _is_allocated(self):               # dat.py:141-143: device storage is created on first use
        return self._dev is not None

    @property
    def _data(self):
        return self._host

    # -- metadata
    @property
    def dataset(self):
        return self._dataset

    @property
    def dim(self):
        return self._dataset.dim

    @property
    def cdim(self):
        return self._dataset.cdim

    @property
    def dtype(self):
        return self._host.dtype

    @property
    def shape(self):
        return self._host.shape

    @property
    def nbytes(self):
        return self.dtype.itemsize * self.dataset.size * self.cdim

    # -- data access (dat.py:134-250)
    @property
    def data(self):
        self.halo_valid = False
        return self._host_rw()[:self.dataset.size]

    @property
    def data_with_halos(self):
        self.global_to_local_begin(RW)
        self.global_to_local_end(RW)
        self.halo_valid = False
        return self._host_rw()

    @property
    def data_ro(self):
        v = self._to_host()[:self.dataset.size].view()
        v.setflags(write=False)
        return v

    @property
    def data_ro_with_halos(self):
        self.global_to_local_begin(READ)
        self.global_to_local_end(READ)
        v = self._to_host().view()
        v.setflags(write=False)
        return v

    def zero(self, subset=None):      # dat.py:297-311
        if subset is not None:
            if subset.superset != self.dataset.set:
                raise MapValueError("The subset and dataset are incompatible")
            # bc.zero(r) (firedrake/bcs.py:192-221): a direct loop over the subset's owned entities
            from .parloop import par_loop
            par_loop(self._kernel("zero"), subset, self(WRITE))
            return
        if self._dev is not None:
            if _capture_log is not None:
                _capture_log[1][id(self)] = self
            self._dev.zero()
            self._dev_valid = True
            self._host_valid = False
            self.dat_version += 1
        else:
            self._host[...] = 0
            self.dat_version += 1
        self.halo_valid = True      # zero everywhere, halos included

    def copy(self, other, subset=None):           # dat.py:313-336
        if other is self:
            return
        if subset is None:
            if self._dev is not None and self._dev_valid and _lib.gpu_available() and self._host.nbytes == other._host.nbytes:
                _lib.call("fd_memcpy_d2d", other._dev_ptr(True), self._dev_ptr(False), self._host.nbytes, None)
            else:
                other._host_rw()[...] = self._to_host()
            other.halo_valid = self.halo_valid
            return
        if subset.superset != self.dataset.set:
            raise MapValueError("The subset and dataset are incompatible")
        from .parloop import par_loop
        par_loop(self._kernel("copy"), subset, self(READ), other(WRITE))

    def assign(self, values):
        self._host_rw()[...] = values

    def __call__(self, access, path=None):
        from .parloop import DatLegacyArg
        if configuration["type_check"] and path is not None and path.toset != self.dataset.set:
            raise MapValueError("To Set of Map does not match Set of Dat.")
        return DatLegacyArg(self, path, access)

    # -- halo state machine (dat.py:622-711)
    def global_to_local_begin(self, access_mode):
        halo = self.dataset.set.halo
        if halo is None or self._halo_frozen:
            return
        if not self.halo_valid and access_mode in (READ, RW):
            halo.global_to_local_begin(self, WRITE)
        elif access_mode in (INC, MIN, MAX):
            halo.fill_ghosts(self, access_mode)      # dat.py:631-636

    def global_to_local_end(self, access_mode):
        halo = self.dataset.set.halo
        if halo is None or self._halo_frozen:
            return
        if not self.halo_valid and access_mode in (READ, RW):
            halo.global_to_local_end(self, WRITE)
            self.halo_valid = True
        elif access_mode in (INC, MIN, MAX):
            self.halo_valid = False

    def local_to_global_begin(self, insert_mode):
        halo = self.dataset.set.halo
        if halo is None or self._halo_frozen:
            return
        halo.local_to_global_begin(self, insert_mode)

    def local_to_global_end(self, insert_mode):
        halo = self.dataset.set.halo
        if halo is None or self._halo_frozen:
            return
        halo.local_to_global_end(self, insert_mode)
        self.halo_valid = False

    @contextmanager
    def frozen_halo(self, access_mode):
        """dat.py:680-711, 1245-1262: suppress per-parloop reverse exchanges; do ONE
        local_to_global when the block exits (OneFormAssembler, assemble.py:1281-1286)."""
        if self._halo_frozen:
            yield
            return
        self.global_to_local_begin(access_mode)
        self.global_to_local_end(access_mode)
        self._halo_frozen = True
        self._frozen_access_mode = access_mode
        try:
            yield
        finally:
            self._halo_frozen = False
            self._frozen_access_mode = None
            if access_mode in (INC, MIN, MAX):
                self.local_to_global_begin(access_mode)
                self.local_to_global_end(access_mode)

    # -- pointwise algebra on the device (pyop2/types/dat.py:354-620): keeps coefficient updates next to the
    #    assemblies (Newton / Runge-Kutta loops) instead of round-tripping through the host
    def axpby(self, a, x: "Dat", b=1.0):
        """self = a*x + b*self  (all rows, halos included)."""
        if x.dataset.total_size != self.dataset.total_size or x.cdim != self.cdim or self.dtype != ScalarType:
            raise DataValueError("axpby needs two float64 Dats of the same shape")
        n = self.dataset.total_size * self.cdim
        _lib.call("fd_dat_axpby", self._dev_ptr(True), ctypes.c_double(a), x._dev_ptr(False), ctypes.c_double(b), n, None)
        self.halo_valid = self.halo_valid and x.halo_valid
        return self

    def assign_dat(self, x: "Dat"):
        return self.axpby(1.0, x, 0.0)

    # -- the reference's Dat arithmetic (pyop2/types/dat.py:354-620): every operation is a DIRECT parloop over the
    #    Dat's own set with a tiny generated kernel (binop_*/iop_*/inner/neg, dat.py:354-470).  Same here: the kernels
    #    are C strings, the loops run through the direct wrapper on the device, so a Newton / Runge-Kutta / Krylov
    #    update never leaves HBM.
    _OPS = {"add": "+", "sub": "-", "mul": "*", "div": "/"}

    def _check_shape(self, other):                # dat.py:349-352
        if other.dataset.dim != self.dataset.dim:
            raise ValueError("Mismatched shapes in operands %s and %s" % (self.dataset.dim, other.dataset.dim))

    @staticmethod
    def _ctype(dt):
        from .codegen import CTYPE
        return CTYPE[np.dtype(dt)]

    def _kernel(self, kind, op=None, globalp=False, other_is_self=False, odtype=None):
        from .kernel import Kernel
        c, ct = self.cdim, self._ctype(self.dtype)
        oct_ = self._ctype(odtype) if odtype is not None else ct
        rhs = "other[0]" if globalp else ("self[i]" if other_is_self else "other[i]")
        loop = f"for (int i = 0; i < {c}; ++i)"
        if kind == "binop":      # dat.py:354-383
            name = f"binop_{op}"
            code = f"static void {name}(const {ct} *self, const {oct_} *other, {ct} *ret) {{ {loop} ret[i] = self[i] {self._OPS[op]} {rhs}; }}"
        elif kind == "iop":      # dat.py:401-436
            name = f"iop_{op}"
            args = f"{ct} *self" + ("" if other_is_self else f", const {oct_} *other")
            code = f"static void {name}({args}) {{ {loop} self[i] = self[i] {self._OPS[op]} {rhs}; }}"
        elif kind == "inner":    # dat.py:454-476
            name = "inner"
            code = f"static void inner(const {ct} *self, const {oct_} *other, {ct} *ret) {{ {loop} ret[0] = ret[0] + self[i]*other[i]; }}"
        elif kind == "neg":      # dat.py:560-582
            name = "neg"
            code = f"static void neg({ct} *other, const {ct} *self) {{ {loop} other[i] = -self[i]; }}"
        elif kind == "axpy":
            name = "axpy"
            code = f"static void axpy({ct} *self, const {ct} *alpha, const {oct_} *other) {{ {loop} self[i] = alpha[0]*other[i] + self[i]; }}"
        elif kind == "copy":
            name = "copy"
            code = f"static void copy(const {ct} *self, {ct} *other) {{ {loop} other[i] = self[i]; }}"
        elif kind == "zero":
            name = "zero"
            code = f"static void zero({ct} *self) {{ {loop} self[i] = 0; }}"
        else:
            raise ValueError(kind)
        # unique per signature: the wrapper cache is keyed by kernel text + name
        return Kernel(code, name)

    def _op(self, other, op):                     # dat.py:385-399
        from .parloop import par_loop
        ret = Dat(self.dataset, None, self.dtype)
        if np.isscalar(other):
            other = Global(1, data=other)
            globalp = True
        else:
            self._check_shape(other)
            globalp = False
        par_loop(self._kernel("binop", op, globalp, odtype=other.dtype), self.dataset.set,
                 self(READ), other(READ), ret(WRITE))
        return ret

    def _iop(self, other, op):                    # dat.py:438-452
        from .parloop import par_loop
        globalp = False
        if np.isscalar(other):
            other = Global(1, data=other)
            globalp = True
        elif isinstance(other, Global):
            globalp = True
        elif other is not self:
            self._check_shape(other)
        args = [self(INC)]
        if other is not self:
            args.append(other(READ))
        par_loop(self._kernel("iop", op, globalp, other is self, odtype=other.dtype), self.dataset.set, *args)
        return self

    def inner(self, other):                       # dat.py:478-492
        """Inner product of the flattened owned data with ``other``."""
        from .parloop import par_loop
        self._check_shape(other)
        ret = Global(1, data=0, dtype=self.dtype)
        par_loop(self._kernel("inner", odtype=other.dtype), self.dataset.set, self(READ), other(READ), ret(INC))
        return ret.data_ro[0]

    @property
    def norm(self):                               # dat.py:494-503
        """L2 norm of the flattened owned data."""
        from math import sqrt
        return sqrt(self.inner(self).real)

    def axpy(self, alpha, other):                 # dat.py:527-546: self = alpha*other + self
        from .parloop import par_loop
        self._check_shape(other)
        if not np.isscalar(alpha):
            raise TypeError("alpha must be a scalar")
        a = Global(1, data=alpha, dtype=self.dtype)
        par_loop(self._kernel("axpy", odtype=other.dtype), self.dataset.set, self(INC), a(READ), other(READ))

    def maxpy(self, scalar, x):                   # dat.py:505-525
        if len(scalar) != len(x):
            raise ValueError("scalar and x must have the same length")
        for alpha_i, x_i in zip(scalar, x):
            self.axpy(alpha_i, x_i)

    def __pos__(self):
        return Dat(self)

    def __neg__(self):                            # dat.py:584-590
        from .parloop import par_loop
        neg = Dat(self.dataset, dtype=self.dtype)
        par_loop(self._kernel("neg"), self.dataset.set, neg(WRITE), self(READ))
        return neg

    def __add__(self, other):
        return self._op(other, "add")

    def __radd__(self, other):
        return self + other

    def __sub__(self, other):
        return self._op(other, "sub")

    def __rsub__(self, other):                    # dat.py:596-603
        ret = -self
        ret += other
        return ret

    def __mul__(self, other):
        return self._op(other, "mul")

    def __rmul__(self, other):
        return self.__mul__(other)

    def __truediv__(self, other):
        return self._op(other, "div")

    __div__ = __truediv__

    def __iadd__(self, other):
        return self._iop(other, "add")

    def __isub__(self, other):
        return self._iop(other, "sub")

    def __imul__(self, other):
        return self._iop(other, "mul")

    def __itruediv__(self, other):
        return self._iop(other, "div")

    __idiv__ = __itruediv__


class DatView(Dat):
    """pyop2/types/dat.py:714-805: a Dat that shows ONE component of a vector/tensor-valued Dat.  It shares the parent's
    storage (host and device) and version / halo state; a kernel sees a single value per node."""

    def __init__(self, dat, index):
        index = (index,) if isinstance(index, (int, np.integer)) else tuple(index)
        if len(index) != len(dat.dim) or not all(0 <= i < d for i, d in zip(index, dat.dim)):
            raise IndexValueError("Can't create DatView with index %s for Dat with shape %s" % (index, dat.dim))
        self.index = tuple(int(i) for i in index)
        self._idx = (slice(None), *self.index)
        self._parent = dat
        self._dataset = dat.dataset
        self.name = "view[%s](%s)" % (self.index, dat.name)
        self._halo_frozen = False
        self._frozen_access_mode = None

    # shared storage and state
    _host = property(lambda self: self._parent._host)
    _dev = property(lambda self: self._parent._dev)
    _host_valid = property(lambda self: self._parent._host_valid)
    _dev_valid = property(lambda self: self._parent._dev_valid)
    dat_version = property(lambda self: self._parent.dat_version)
    dtype = property(lambda self: self._parent.dtype)
    cdim = property(lambda self: 1)
    dim = property(lambda self: (1,))
    shape = property(lambda self: (self._dataset.total_size,))

    @property
    def halo_valid(self):
        return self._parent.halo_valid

    @halo_valid.setter
    def halo_valid(self, value):
        self._parent.halo_valid = value

    def _to_host(self):
        return self._parent._to_host()

    def _host_rw(self):
        return self._parent._host_rw()

    def _dev_ptr(self, write):
        return self._parent._dev_ptr(write)

    def _after_device_write(self):
        self._parent._after_device_write()

    data = property(lambda self: self._parent.data[self._idx])
    data_ro = property(lambda self: self._parent.data_ro[self._idx])
    data_with_halos = property(lambda self: self._parent.data_with_halos[self._idx])
    data_ro_with_halos = property(lambda self: self._parent.data_ro_with_halos[self._idx])
    _data = property(lambda self: self._parent._host[self._idx])

    def zero(self, subset=None):               # dat.py:297-311 on the view's slice
        if subset is not None:
            return Dat.zero(self, subset)
        if self._parent._dev is not None and self._parent._dev_valid and _lib.gpu_available():
            from .parloop import par_loop
            par_loop(self._kernel("zero"), self._dataset.set, self(WRITE))
            if self._dataset.set.total_size > self._dataset.set.size:     # ghosts too: "zero everywhere"
                self._parent._host_rw()[(slice(self._dataset.size, None), *self.index)] = 0
        else:
            self._parent._host_rw()[self._idx] = 0
            self._parent.dat_version += 1
        self.halo_valid = True

    def __call__(self, access, path=None):
        from .parloop import DatLegacyArg
        if configuration["type_check"] and path is not None and path.toset != self._dataset.set:
            raise MapValueError("To Set of Map does not match Set of Dat.")
        return DatLegacyArg(self, path, access)


class MixedDat:
    """pyop2/types/dat.py:861-1243: a bag of Dats (the coefficient vector of a mixed function space).  Built from a
    MixedDataSet / MixedSet / iterable of (Data)Sets, or from an iterable of Dats."""

    def __init__(self, mdset_or_dats):
        if isinstance(mdset_or_dats, str):
            raise DataSetTypeError("MixedDat needs a MixedDataSet, a MixedSet or an iterable of Dats / (Data)Sets")
        if isinstance(mdset_or_dats, MixedDat):
            self._dats = tuple(Dat(d) for d in mdset_or_dats)
        else:
            self._dats = tuple(d if isinstance(d, Dat) else Dat(d) for d in mdset_or_dats)
        if not self._dats:
            raise DataValueError("MixedDat needs at least one Dat")
        if not all(d.dtype == self._dats[0].dtype for d in self._dats):
            raise DataValueError("MixedDat with different dtypes is not supported")
        self.comm = self._dats[0].dataset.set.comm

    split = property(lambda self: self._dats)
    dtype = property(lambda self: self._dats[0].dtype)
    dim = property(lambda self: self.dataset.dim)
    cdim = property(lambda self: self.dataset.cdim)
    dat_version = property(lambda self: sum(d.dat_version for d in self._dats))
    data = property(lambda self: tuple(d.data for d in self._dats))
    data_ro = property(lambda self: tuple(d.data_ro for d in self._dats))
    data_with_halos = property(lambda self: tuple(d.data_with_halos for d in self._dats))
    data_ro_with_halos = property(lambda self: tuple(d.data_ro_with_halos for d in self._dats))
    nbytes = property(lambda self: sum(d.nbytes for d in self._dats))

    @property
    def dataset(self):
        return MixedDataSet(tuple(d.dataset for d in self._dats))

    @property
    def halo_valid(self):
        return all(d.halo_valid for d in self._dats)

    @halo_valid.setter
    def halo_valid(self, val):
        for d in self._dats:
            d.halo_valid = val

    def __getitem__(self, idx):
        return self._dats[idx]

    def __iter__(self):
        return iter(self._dats)

    def __len__(self):
        return len(self._dats)

    def __call__(self, access, path=None):
        from .parloop import MixedDatLegacyArg
        return MixedDatLegacyArg(self, path, access)

    def zero(self, subset=None):      # dat.py:1028-1036
        if subset is not None:
            raise NotImplementedError("Subsets of mixed sets not implemented")
        for d in self._dats:
            d.zero()

    def copy(self, other, subset=None):
        if subset is not None:
            raise NotImplementedError("MixedDat.copy with a Subset is not supported")
        for s_, o in zip(self, other):
            s_.copy(o)

    def global_to_local_begin(self, access_mode):
        for d in self._dats:
            d.global_to_local_begin(access_mode)

    def global_to_local_end(self, access_mode):
        for d in self._dats:
            d.global_to_local_end(access_mode)

    def local_to_global_begin(self, insert_mode):
        for d in self._dats:
            d.local_to_global_begin(insert_mode)

    def local_to_global_end(self, insert_mode):
        for d in self._dats:
            d.local_to_global_end(insert_mode)

    def __hash__(self):
        return hash(self._dats)

    def __eq__(self, other):               # dat.py:1074-1082
        return type(self) is type(other) and all(a is b for a, b in zip(self._dats, other._dats)) \
            and len(self._dats) == len(other._dats)

    def __ne__(self, other):
        return not self.__eq__(other)

    def __str__(self):
        return "OP2 MixedDat composed of Dats: %s" % (self._dats,)

    def __repr__(self):
        return "MixedDat(%r)" % (self._dats,)

    # -- arithmetic: component-wise (dat.py:1090-1198)
    def inner(self, other):
        return sum(s_.inner(o) for s_, o in zip(self, other))

    @property
    def norm(self):
        from math import sqrt
        return sqrt(self.inner(self).real)

    def axpy(self, alpha, other):
        for s_, o in zip(self, other):
            s_.axpy(alpha, o)

    def _op(self, other, op):
        if np.isscalar(other):
            return MixedDat([getattr(s_, op)(other) for s_ in self])
        return MixedDat([getattr(s_, op)(o) for s_, o in zip(self, other)])

    def _iop(self, other, op):
        if np.isscalar(other):
            for s_ in self:
                getattr(s_, op)(other)
        else:
            for s_, o in zip(self, other):
                getattr(s_, op)(o)
        return self

    def __pos__(self):
        return MixedDat([+d for d in self])

    def __neg__(self):
        return MixedDat([-d for d in self])

    def __add__(self, other):
        return self._op(other, "__add__")

    def __radd__(self, other):
        return self._op(other, "__radd__")

    def __sub__(self, other):
        return self._op(other, "__sub__")

    def __rsub__(self, other):
        return self._op(other, "__rsub__")

    def __mul__(self, other):
        return self._op(other, "__mul__")

    def __rmul__(self, other):
        return self._op(other, "__rmul__")

    def __truediv__(self, other):
        return self._op(other, "__truediv__")

    def __iadd__(self, other):
        return self._iop(other, "__iadd__")

    def __isub__(self, other):
        return self._iop(other, "__isub__")

    def __imul__(self, other):
        return self._iop(other, "__imul__")

    def __itruediv__(self, other):
        return self._iop(other, "__itruediv__")


class Global(_Mirrored):
    """pyop2/types/glob.py:21-480."""

    def __init__(self, dim, data=None, dtype=None, name=None, comm=None):
        _check_name(name)
        if isinstance(dim, Global):
            data, dtype, dim = dim.data_ro.copy(), dim.dtype, dim.dim
        if isinstance(dim, (int, np.integer)):
            dim = (int(dim),)
        try:                                                       # glob.py:29 (as_tuple(dim, int))
            if isinstance(dim, str) or not all(isinstance(d, (int, np.integer)) for d in dim):
                raise TypeError
            self._dim = tuple(int(d) for d in dim)
        except TypeError:
            raise DimTypeError(f"Global dim must be an int or a tuple of ints, got {dim!r}")
        n = int(np.prod(self._dim))
        try:                                                       # utils.verify_reshape: bad dtype / values / length
            dt = np.dtype(dtype) if dtype is not None else (np.asarray(data).dtype if data is not None else ScalarType)
            if data is None:
                host = np.zeros(self._dim, dtype=dt)
            else:
                a = np.asarray(data, dtype=dt)
                if a.size == 1 and n != 1:
                    host = np.full(self._dim, a, dtype=dt)
                else:
                    host = a.reshape(self._dim).copy()
        except (TypeError, ValueError):
            raise DataValueError("Invalid data for a Global of dim %s: %r" % (self._dim, data))
        self._init_storage(host)
        self.name = name or f"global_#x{id(self):x}"
        self.comm = comm

    @property
    def dim(self):
        return self._dim

    @property
    def cdim(self):
        return int(np.prod(self._dim))

    @property
    def dtype(self):
        return self._host.dtype

    @property
    def data(self):
        return self._host_rw()

    @data.setter
    def data(self, value):                 # glob.py:160-163 (verify_reshape: wrong length -> DataValueError)
        try:
            v = np.asarray(value, dtype=self.dtype)
            if v.size != 1 and v.size != self._host.size:
                raise ValueError
            self._host_rw()[...] = v.reshape(self._host.shape) if v.size == self._host.size else v
        except (TypeError, ValueError):
            raise DataValueError("Invalid data: expected %d values, got %r" % (self._host.size, value))

    @property
    def data_ro(self):
        v = self._to_host().view()
        v.setflags(write=False)
        return v

    def zero(self):
        self._host_rw()[...] = 0

    _modes = (READ, INC, MIN, MAX)            # glob.py:91

    def __call__(self, access, map_=None):
        from .parloop import GlobalLegacyArg
        if access not in self._modes:
            raise ModeValueError("Global arguments may be accessed with READ, INC, MIN or MAX")
        assert map_ is None
        return GlobalLegacyArg(self, access)

    # glob.py:100-125, 292-298: a Global is a one-member bag of itself
    def __getitem__(self, idx):
        if idx != 0:
            raise IndexValueError("Can only extract component 0 from %r" % self)
        return self

    split = property(lambda self: (self,))
    shape = property(lambda self: self._dim)

    def __iter__(self):
        yield self

    def __len__(self):
        return 1

    def __str__(self):
        return "OP2 Global Argument: %s with dim %s and value %s" % (self.name, self._dim, self._to_host())

    def __repr__(self):
        return "Global(%r, %r, %r, %r)" % (self._dim, self._to_host(), self.dtype, self.name)


Constant = Global


class MixedSet:
    """pyop2/types/set.py:546-660: an ordered bag of Sets (the node sets of a mixed function space)."""

    def __init__(self, sets):
        sets = tuple(sets.split if isinstance(sets, MixedSet) else sets)
        if not sets or not all(isinstance(s, Set) for s in sets):
            raise SetTypeError("All sets of a MixedSet must be of type Set")
        if len({s._extruded for s in sets}) != 1:
            raise AssertionError("All components of a MixedSet must have the same extrusion")      # set.py:552-556
        if len({s.layers for s in sets}) != 1:
            raise AssertionError("All components of a MixedSet must have the same number of layers")
        self._sets = sets
        self.comm = sets[0].comm

    split = property(lambda self: self._sets)
    name = property(lambda self: tuple(s.name for s in self._sets))          # set.py:619-622
    layers = property(lambda self: self._sets[0].layers)                      # set.py:639-642

    @property
    def halo(self):                        # set.py:624-628
        halos = tuple(s.halo for s in self._sets)
        return halos if any(halos) else None

    _extruded = property(lambda self: self._sets[0]._extruded)
    core_size = property(lambda self: sum(s.core_size for s in self._sets))
    size = property(lambda self: sum(s.size for s in self._sets))
    total_size = property(lambda self: sum(s.total_size for s in self._sets))
    sizes = property(lambda self: (self.core_size, self.size, self.total_size))
    superset = property(lambda self: self)

    def __getitem__(self, idx):
        return self._sets[idx]

    def __iter__(self):
        return iter(self._sets)

    def __len__(self):
        return len(self._sets)

    def __pow__(self, e):
        return MixedDataSet(self._sets, e)

    def __eq__(self, o):
        return type(self) is type(o) and self._sets == o._sets

    def __ne__(self, o):
        return not self == o

    def __hash__(self):
        return hash(self._sets)

    def __str__(self):
        return "OP2 MixedSet composed of Sets: %s" % (self._sets,)

    def __repr__(self):
        return "MixedSet(%r)" % (self._sets,)


class _ShapeOnly:
    """Stands in for the host values of a Map that exists on the device only."""

    def __init__(self, shape):
        self.shape = tuple(shape)

    def __len__(self):
        return self.shape[0]


# ---- maps --------------------------------------------------------------------------------
class Map:
    """pyop2/types/map.py:17-110: int array (iterset.total_size, arity) of IntType;
    ``offset`` gives the extruded per-entry node stride (map.py:46-53)."""

    VALUE_UNDEFINED = -1

    def __init__(self, iterset, toset, arity, values=None, name=None, offset=None, offset_quotient=None):
        if not isinstance(iterset, Set) or not isinstance(toset, Set):
            raise SetTypeError("Map iterset/toset must be Sets")
        if not isinstance(arity, (int, np.integer)) or isinstance(arity, bool):
            raise ArityTypeError("Map arity must be an int")          # map.py:34 (validate_type)
        _check_name(name)
        self._iterset = iterset
        self._toset = toset
        self._arity = int(arity)
        if values is None:
            self._values = np.zeros((0, self._arity), dtype=IntType)
        else:
            try:
                v = np.asarray(values)
                self._values = np.ascontiguousarray(v.astype(IntType, copy=False).reshape(iterset.total_size, self._arity))
            except (TypeError, ValueError):
                raise DataValueError("Invalid data: expected %d integer values, got %r" % (iterset.total_size * self._arity, values))
        self.name = name or f"map_#x{id(self):x}"
        self._offset = None if offset is None else tuple(int(o) for o in offset)
        if offset_quotient is None or len(offset_quotient) == 0:        # map.py:50-53
            self._offset_quotient = None
        else:
            if len(offset_quotient) != self._arity:
                raise DataValueError("offset_quotient must have one entry per map entry")
            self._offset_quotient = tuple(int(o) for o in offset_quotient)
        self._dev = None
        self._plans = {}

    iterset = property(lambda self: self._iterset)
    toset = property(lambda self: self._toset)
    arity = property(lambda self: self._arity)
    offset = property(lambda self: self._offset)
    offset_quotient = property(lambda self: self._offset_quotient)

    @property
    def values(self):
        return self._values[:self.iterset.size]

    @property
    def values_with_halo(self):
        return self._values

    @property
    def arities(self):
        return (self._arity,)

    arange = property(lambda self: (0, self._arity))      # map.py:115-118
    split = property(lambda self: (self,))

    def __len__(self):
        return 1

    def __iter__(self):
        yield self

    def __getitem__(self, idx):
        if idx != 0:
            raise IndexValueError("Can only extract component 0 from %r" % self)
        return self

    def __le__(self, o):                   # map.py:160-162
        return self == o

    def __str__(self):
        return "OP2 Map: %s from (%s) to (%s) with arity %s" % (self.name, self._iterset, self._toset, self._arity)

    def __repr__(self):
        return "Map(%r, %r, %r, None, %r, %r, %r)" % (self._iterset, self._toset, self._arity, self.name, self._offset,
                                                      self._offset_quotient)

    def _base(self):
        return self

    def _dev_values(self):
        if self._dev is None:
            self._dev = DeviceBuffer.from_numpy(self._values)
        return self._dev.ptr

    def derived(self, key, build):
        """A non-extruded, unsubsetted Map over a VIRTUAL iteration space derived from this one -- the rows of a Subset's
        entities, or one row ``map + offset*layer`` per (column, layer) cell of an extruded set -- built once by
        ``build()`` (an (n, arity) int32 array) and cached: what the staged wrapper's plans are made of on subsets and
        extruded sets (codegen.staged_eligible)."""
        d = self._derived.get(key) if hasattr(self, "_derived") else None
        if d is None:
            if not hasattr(self, "_derived"):
                self._derived = {}
            vals = np.ascontiguousarray(build(), dtype=IntType)
            # (interior facets of an extruded set: a row holds the nodes of BOTH stacked cells, twice the arity)
            d = Map(Set(len(vals), "virtual_" + self.iterset.name), self.toset, vals.shape[1] if vals.ndim == 2 else self.arity, vals,
                    self.name + "_derived")
            self._derived[key] = d
        return d

    def derived_dev(self, key, nrows, build_dev):
        """Like ``derived`` for rows gathered ON THE DEVICE (``build_dev()`` returns a DeviceBuffer of nrows x arity int32):
        the derived Map has no host values."""
        if not hasattr(self, "_derived"):
            self._derived = {}
        d = self._derived.get(key)
        if d is None:
            d = Map.__new__(Map)
            d._iterset, d._toset, d._arity = Set(int(nrows), "ordered_" + self.iterset.name), self.toset, self.arity
            d._values = _ShapeOnly((int(nrows), self.arity))
            d.name, d._offset, d._offset_quotient, d._plans = self.name + "_ordered", None, None, {}
            d._dev = build_dev()
            self._derived[key] = d
        return d

    def plan(self, start, end, epb, blocks=None, lane_threads=0):
        """Cached block-localisation plan for [start, end) (include/fdhip.h: fd_plan_create[_blocks]).
        ``blocks``: optional int32 array of block boundaries (entity offsets, first = start, last = end)."""
        key = (int(start), int(end), int(epb) if blocks is None else ("blocks", len(blocks), int(np.asarray(blocks).sum() % 2147483647)),
               int(lane_threads))
        p = self._plans.get(key)
        if p is None:
            p = Plan(self, int(start), int(end), int(epb), blocks, lane_threads=lane_threads)
            self._plans[key] = p
        return p


class PermutedMap(Map):
    """pyop2/types/map.py:113-156: same values, entries visited as map[n][perm[i]]."""

    def __init__(self, map_, permutation):
        if not isinstance(map_, Map):
            raise SetTypeError("PermutedMap needs a Map")
        self.map_ = map_
        self.permutation = np.asarray(permutation, dtype=IntType)
        if sorted(self.permutation.tolist()) != list(range(map_.arity)):
            raise DataValueError("permutation must be a permutation of range(arity)")
        self.name = map_.name + "_perm"

    def __getattr__(self, name):
        return getattr(self.map_, name)

    def _base(self):
        return self.map_._base()


class ComposedMap(Map):
    """pyop2/types/map.py:206-270: ``local[i] = global[maps_[0][maps_[1][maps_[2][...]]][i]]`` with the inner maps of
    arity 1.  The reference indexes through the chain inside the generated wrapper (builder.py:179-198); here the
    chain is composed once on the host into an ordinary (iterset, arity) table -- a lossless re-encoding, so the
    wrapper kernels see a plain Map.  Undefined (negative) intermediate entries give a row of VALUE_UNDEFINED."""

    def __init__(self, *maps_, name=None):
        flat = []
        for m in maps_:
            if not isinstance(m, Map):
                raise TypeError("All maps must be Map instances")
            flat.extend(m.maps_ if isinstance(m, ComposedMap) else [m])
        for tomap, frommap in zip(flat[:-1], flat[1:]):
            if tomap.iterset.superset is not frommap.toset.superset and tomap.iterset is not frommap.toset:
                raise MapValueError("tomap.iterset must match frommap.toset")
            if frommap.arity != 1:
                raise MapValueError("frommap.arity must be 1")
        self.maps_ = tuple(flat)
        first = flat[0]
        rows = first.values_with_halo
        if isinstance(first, PermutedMap):
            rows = first.map_.values_with_halo[:, first.permutation]
        idx = None
        for m in reversed(flat[1:]):
            v = m.values_with_halo[:, 0]
            idx = v if idx is None else np.where(idx >= 0, v[np.maximum(idx, 0)], -1)
        vals = np.where((idx >= 0)[:, None], rows[np.maximum(idx, 0)], self.VALUE_UNDEFINED).astype(IntType)
        super().__init__(flat[-1].iterset, first.toset, first.arity, vals, name or f"cmap_{id(self):x}",
                         offset=first.offset)


class MixedMap:
    """pyop2/types/map.py:330-470: one Map per component of a mixed space, all on the same iteration set
    (entries may be None for components an argument does not touch)."""

    def __init__(self, maps):
        if isinstance(maps, str):
            raise MapTypeError("MixedMap needs an iterable of Maps")
        maps = tuple(maps.split if isinstance(maps, MixedMap) else maps)
        if not maps or not all(m is None or isinstance(m, Map) for m in maps):
            raise MapTypeError("MixedMap needs an iterable of Maps")
        present = [m for m in maps if m is not None]
        if not present:
            raise TypeError("Don't know how to make a MixedMap of no Maps")
        if any(m.iterset is not present[0].iterset for m in present):
            raise MapValueError("All maps in a MixedMap need to share the same iterset")       # map.py:378-385
        self._maps = maps

    split = property(lambda self: self._maps)
    iterset = property(lambda self: next(m for m in self._maps if m is not None).iterset)
    toset = property(lambda self: MixedSet(tuple(m.toset for m in self._maps)))
    arity = property(lambda self: sum(m.arity for m in self._maps))
    arities = property(lambda self: tuple(m.arity for m in self._maps))
    values = property(lambda self: tuple(m.values for m in self._maps))
    values_with_halo = property(lambda self: tuple(m.values_with_halo for m in self._maps))
    offset = property(lambda self: tuple(0 if m is None else m.offset for m in self._maps))
    offset_quotient = property(lambda self: tuple(0 if m is None else m.offset_quotient for m in self._maps))
    name = property(lambda self: tuple(m.name for m in self._maps))            # map.py:434-437
    arange = property(lambda self: (0,) + tuple(np.cumsum(self.arities)))      # map.py:411-414

    def __eq__(self, o):
        return type(self) is type(o) and len(o) == len(self) and all(a is b for a, b in zip(self._maps, o._maps))

    def __ne__(self, o):
        return not self == o

    def __hash__(self):
        return hash(tuple(id(m) for m in self._maps))

    def __le__(self, o):                   # map.py:458-460
        return self == o or all(m <= om for m, om in zip(self, o))

    def __str__(self):
        return "OP2 MixedMap composed of Maps: %s" % (self._maps,)

    def __iter__(self):
        return iter(self._maps)

    def __len__(self):
        return len(self._maps)

    def __getitem__(self, idx):
        return self._maps[idx]

    def __repr__(self):
        return "MixedMap(%r)" % (self._maps,)


class Plan:
    """Python handle on an fd_plan_t."""

    def __init__(self, map_, start, end, epb, blocks=None, arity=None, lane_threads=0):
        """``map_``: a Map, or a raw device pointer to an int32 (n, arity) array (then pass ``arity``).
        ``lane_threads`` > 0: local-map rows in lane order (include/fdhip.h: fd_plan_set_lane_order)."""
        h = ctypes.c_void_p()
        if isinstance(map_, Map):
            dev, arity = map_._dev_values(), map_.arity
        else:
            dev = int(map_)
        self.arity = arity
        if blocks is None:
            _lib.call("fd_plan_create", dev, arity, start, end, epb, None, ctypes.byref(h))
        else:
            bl = np.ascontiguousarray(blocks, dtype=np.int32)
            assert bl[0] == start and bl[-1] == end
            _lib.call("fd_plan_create_blocks", dev, arity, bl.ctypes.data, len(bl) - 1, None, ctypes.byref(h))
        self.h = h.value
        self.lane_threads = int(lane_threads)
        if self.lane_threads > 0:
            _lib.call("fd_plan_set_lane_order", self.h, self.lane_threads, None)
        bs, me = ctypes.c_void_p(), ctypes.c_int32()
        _lib.call("fd_plan_block_starts", self.h, ctypes.byref(bs), ctypes.byref(me))
        self.bstart, epb = bs.value, me.value
        nb, mx, ll = ctypes.c_int32(), ctypes.c_int32(), ctypes.c_int64()
        _lib.call("fd_plan_info", self.h, ctypes.byref(nb), ctypes.byref(mx), ctypes.byref(ll))
        self.nblocks, self.max_nd, self.list_len = nb.value, mx.value, ll.value
        a, b, c = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_void_p()
        _lib.call("fd_plan_arrays", self.h, ctypes.byref(a), ctypes.byref(b), ctypes.byref(c))
        self.blkoff, self.list, self.lmap = a.value, b.value, c.value
        self.start, self.end, self.epb = start, end, epb

    def download(self):
        """(block_offsets, node_list, local_map) as numpy arrays -- for tests/diagnostics."""
        blk = np.empty(self.nblocks + 1, dtype=np.int32)
        lst = np.empty(self.list_len, dtype=np.int32)
        lm = np.empty((self.end - self.start, self.arity), dtype=np.uint16)
        for arr, p in ((blk, self.blkoff), (lst, self.list), (lm, self.lmap)):
            if arr.nbytes:
                _lib.call("fd_memcpy_d2h", arr.ctypes.data, p, arr.nbytes, None)
        return blk, lst, lm

    def __del__(self):
        try:
            if self.h:
                _lib.load().fd_plan_free(self.h)
        except Exception:
            pass


# ---- sparsity / matrix -------------------------------------------------------------------
class Sparsity:
    """pyop2/types/mat.py:27-291.  ``maps_and_regions`` = [(rmap, cmap, iteration_regions)];
    the CSR pattern is built natively on the device (fd_csr_from_maps) instead of walking a
    PETSc MATPREALLOCATOR (pyop2/sparsity.pyx:105-159)."""

    # non-nested: "one block, itself" (mat.py:87-99, 741-768) WITHOUT storing the list -- ``self._blocks = [[self]]`` is a reference cycle,
    # and a Sparsity / Mat in a cycle keeps its device arrays (and, through its maps, their plans and derived orders) until the cycle
    # collector runs instead of until the last reference goes (tests/test_gpu_leaks.py)
    _nested_blocks = None

    @property
    def _blocks(self):
        return self._nested_blocks if self._nested_blocks is not None else [[self]]

    @_blocks.setter
    def _blocks(self, rows):
        self._nested_blocks = None if (len(rows) == 1 and len(rows[0]) == 1 and rows[0][0] is self) else rows

    def __init__(self, dsets, maps_and_regions, name=None, nest=None, block_sparse=None, diagonal_block=True):
        if isinstance(dsets, (Set, DataSet, MixedSet, MixedDataSet)):
            dsets = (dsets, dsets)
        if len(dsets) != 2:
            raise RuntimeError(f"dsets must be a tuple of two DataSets: got {dsets}")
        dsets = tuple(MixedDataSet(d) if isinstance(d, MixedSet) else (DataSet(d) if isinstance(d, Set) else d)
                      for d in dsets)
        for d in dsets:
            if not isinstance(d, (DataSet, MixedDataSet)):
                raise DataSetTypeError("All data sets must be of type DataSet, not type %r" % type(d))     # mat.py:119-121
        _check_name(name)
        self.name = name or f"sparsity_#x{id(self):x}"
        self._built = False
        self._block_sparse = block_sparse
        self._diagonal_block = diagonal_block
        if isinstance(maps_and_regions, (list, tuple)):
            maps_and_regions = {(0, 0): maps_and_regions}          # short-hand for a single block (mat.py:125-127)
        elif not isinstance(maps_and_regions, dict):
            raise TypeError(f"maps_and_regions must be dict or Sequence: got {type(maps_and_regions)}")
        # mat.py:128-160: validate, de-duplicate and order the (rmap, cmap, regions) triples of every block
        processed = {(i, j): () for i in range(len(dsets[0])) for j in range(len(dsets[1]))}
        for (i, j), val in maps_and_regions.items():
            if i >= len(dsets[0]) or j >= len(dsets[1]):
                raise RuntimeError(f"(i, j) must be < {(len(dsets[0]), len(dsets[1]))}: got {(i, j)}")
            seen = []
            for entry in val:
                if isinstance(entry, Map):
                    entry = (entry, entry, None)
                rmap, cmap = entry[0], entry[1]
                regions = entry[2] if len(entry) > 2 else None
                for m in (rmap, cmap):
                    if not isinstance(m, Map):
                        raise MapTypeError("All maps must be of type map, not type %r" % type(m))
                    if not isinstance(m, ComposedMap) and len(m.values_with_halo) == 0 and m.iterset.total_size > 0:
                        raise MapValueError("Unpopulated map values when trying to build sparsity.")
                if rmap.toset is not dsets[0][i].set or cmap.toset is not dsets[1][j].set:
                    raise RuntimeError("Map toset must be the same as DataSet set")
                if rmap.iterset.superset is not cmap.iterset.superset:
                    raise RuntimeError("Iterset of both maps in a pair must be the same")
                regions = (ALL,) if regions is None else tuple(sorted(regions))
                if not any(t[0] is rmap and t[1] is cmap and t[2] == regions for t in seen):
                    seen.append((rmap, cmap, regions))
            # a deterministic order whatever the order the pairs were given in
            processed[(i, j)] = tuple(sorted(seen, key=lambda t: (t[0].name, t[1].name, tuple(int(r) for r in t[2]))))
        self._maps_and_regions = dict(sorted(processed.items()))
        self._dsets = dsets
        if any(isinstance(d, MixedDataSet) for d in dsets):
            # mixed spaces: one Sparsity per block, each built on its own (mat.py:87-99, MATNEST)
            if nest is False:
                raise NotImplementedError("monolithic (nest=False) sparsities over mixed sets are out of scope")
            self._nested = True
            same = dsets[0] is dsets[1] or dsets[0] == dsets[1]
            self._blocks = [[Sparsity((rds, cds), list(self._maps_and_regions[(i, j)]), block_sparse=block_sparse,
                                      diagonal_block=(same and i == j))
                             for j, cds in enumerate(dsets[1])] for i, rds in enumerate(dsets[0])]
            self._pairs = []
            self._has_diagonal = False
            return
        self._nested = False
        self._blocks = [[self]]
        self._pairs = list(self._maps_and_regions[(0, 0)])
        self._has_diagonal = diagonal_block and self._dsets[0].set is self._dsets[1].set

    @property
    def rcmaps(self):                  # mat.py:195-197
        return {key: [(r, c) for r, c, _ in val] for key, val in self._maps_and_regions.items()}

    @property
    def iteration_regions(self):       # mat.py:199-201
        return {key: [reg for _, _, reg in val] for key, val in self._maps_and_regions.items()}

    def __str__(self):
        return "OP2 Sparsity: dsets %s, maps_and_regions %s, name %s, nested %s, block_sparse %s, diagonal_block %s" % \
            (self._dsets, self._maps_and_regions, self.name, self._nested, self._block_sparse, self._diagonal_block)

    def __repr__(self):
        return "Sparsity(%r, %r, name=%r, nested=%r, block_sparse=%r, diagonal_block=%r)" % \
            (self.dsets, self._maps_and_regions, self.name, self._nested, self._block_sparse, self._diagonal_block)

    def __getitem__(self, idx):        # mat.py:180-187
        try:
            i, j = idx
            return self._blocks[i][j]
        except TypeError:
            return self._blocks[idx]

    def __iter__(self):                # blocks in row-major order (mat.py:216-220)
        for row in self._blocks:
            yield from row

    nested = property(lambda self: self._nested)

    dsets = property(lambda self: self._dsets)

    @property
    def dims(self):                    # mat.py:203-212: (rows per row-set entry, cols per col-set entry) of every block
        return tuple(tuple((r.cdim, c.cdim) for c in self._dsets[1]) for r in self._dsets[0])

    @property
    def shape(self):
        return (len(self._dsets[0]), len(self._dsets[1]))

    def _build(self):
        if self._built:
            return
        if self._nested:
            for blk in self:
                blk._build()
            self._built = True
            return
        _lib.require_gpu()
        rset, cset = self._dsets
        # one pattern contribution per (map pair, iteration region): sparsity.pyx:291-305
        pairs = [(r, c, reg) for (r, c, regions) in self._pairs
                 for reg in (regions if r.iterset._extruded else (ALL,))]
        n = len(pairs)
        VP = ctypes.c_void_p
        rm, cm = (VP * n)(), (VP * n)()
        ro, co, rq, cq, lay = (VP * n)(), (VP * n)(), (VP * n)(), (VP * n)(), (VP * n)()
        nent, ra, ca, nl, region, periodic = ((ctypes.c_int32 * n)() for _ in range(6))
        keep = []

        def host_ints(x):
            a = np.asarray(x, dtype=np.int32)
            keep.append(a)
            return a.ctypes.data

        for k, (r, c, reg) in enumerate(pairs):
            rm[k], cm[k] = r._base()._dev_values(), c._base()._dev_values()
            it = r.iterset
            # The reference walks the owned entities only (sparsity.pyx: set_size = iterset.size) and lets
            # PETSc stash off-process rows; with owner-computes-rows the ghost entities contribute to owned
            # rows too, so the pattern is built over owned + ghost entities.
            nent[k] = it.total_size if not isinstance(it, Subset) else it.superset.total_size
            ra[k], ca[k] = r.arity, c.arity
            region[k] = int(reg)
            if it._extruded:
                if it.constant_layers:
                    nl[k] = it.layers - 1
                else:
                    la = it.layers_array                       # per-entity [bottom, top): the longest column sets the slots
                    nl[k] = int((la[:, 1] - 1 - la[:, 0]).max()) if len(la) else 0
                    lay[k] = it._layers_dev()
                if r.offset is None or c.offset is None:
                    raise MapValueError("maps of an extruded iteration set need offsets to build a sparsity")
                ro[k], co[k] = host_ints(r.offset), host_ints(c.offset)
                periodic[k] = int(bool(it._extruded_periodic))
                if r.offset_quotient is not None:
                    rq[k] = host_ints(r.offset_quotient)
                if c.offset_quotient is not None:
                    cq[k] = host_ints(c.offset_quotient)
            else:
                nl[k] = 0
        rp, ci, nnz = VP(), VP(), ctypes.c_int64()
        _lib.call("fd_csr_from_maps_ex", rset.set.total_size, cset.set.total_size, int(self._has_diagonal), n,
                  rm, cm, nent, ra, ca, nl, ro, co, region, periodic, rq, cq, lay,
                  ctypes.byref(rp), ctypes.byref(ci), ctypes.byref(nnz), None)
        self._node_rowptr = DeviceBuffer.wrap(rp.value, (rset.set.total_size + 1) * _lib.NNZ_BYTES)
        self._node_colidx = DeviceBuffer.wrap(ci.value, max(nnz.value, 1) * 4)
        self._node_nnz = nnz.value
        rbs, cbs = rset.cdim, cset.cdim
        if rbs == 1 and cbs == 1:
            self._rowptr, self._colidx, self._nnz = self._node_rowptr, self._node_colidx, nnz.value
        else:
            rp2, ci2 = VP(), VP()
            _lib.call("fd_csr_expand_blocks", rset.set.total_size, rp.value, ci.value, rbs, cbs,
                      ctypes.byref(rp2), ctypes.byref(ci2), None)
            self._nnz = nnz.value * rbs * cbs
            self._rowptr = DeviceBuffer.wrap(rp2.value, (rset.set.total_size * rbs + 1) * _lib.NNZ_BYTES)
            self._colidx = DeviceBuffer.wrap(ci2.value, max(self._nnz, 1) * 4)
        self._built = True
        self._elem_tables = {}

    @property
    def nrows(self):
        return self._dsets[0].set.total_size * self._dsets[0].cdim

    @property
    def ncols(self):
        return self._dsets[1].set.total_size * self._dsets[1].cdim

    @property
    def nz(self):
        self._build()
        return self._nnz

    @property
    def rowptr(self):
        self._build()
        return self._rowptr.download(_lib.NNZ_DTYPE, (self.nrows + 1,))

    def _node_rowptr_host(self):
        """Host copy of the node-level row starts (read-only; the pattern of a built Sparsity never changes): the plan builders look
        at it several times per plan (block cuts, accumulator sizes, longest row)."""
        self._build()
        h = self.__dict__.get("_node_rowptr_h")
        if h is None:
            h = self._node_rowptr.download(_lib.NNZ_DTYPE, (self.dsets[0].set.total_size + 1,))
            h.setflags(write=False)
            self.__dict__["_node_rowptr_h"] = h
        return h

    def _max_node_rowlen(self):
        """Longest node row (cached like the host row starts)."""
        m = self.__dict__.get("_max_node_rowlen_v")
        if m is None:
            rp = self._node_rowptr_host()
            m = self.__dict__["_max_node_rowlen_v"] = int(np.diff(rp).max()) if len(rp) > 1 else 0
        return m

    @property
    def colidx(self):
        self._build()
        return self._colidx.download(np.int32, (self._nnz,))

    @property
    def nnz(self):
        """Non-zeroes per OWNED scalar row in the diagonal portion of the local submatrix: ``d_nnz`` of
        MatMPIAIJSetPreallocation (mat.py:254-262)."""
        return self.mpiaij_split().row_counts()[0]

    @property
    def onnz(self):
        """Non-zeroes per owned scalar row in the off-diagonal portion (columns owned by other ranks): ``o_nnz``
        (mat.py:264-271)."""
        return self.mpiaij_split().row_counts()[1]

    def mpiaij_split(self, col_global=None):
        """The owned rows of the scalar CSR as the two sequential blocks of an MPIAIJ matrix (fd_csr_split_mpiaij): diagonal
        block = columns owned here (local indices), off-diagonal block = ghost columns, numbered through ``col_global``
        (int32 array over the local columns: the column lgmap of pyop2/types/dataset.py:120-164; None keeps local numbers).
        Cached per ``col_global`` object; ``Mat.mpiaij_values`` fills the matching value arrays."""
        self._build()
        cache = self.__dict__.setdefault("_mpiaij", {})
        key = id(col_global)
        hit = cache.get(key)
        if hit is None or hit[0] is not col_global:
            hit = (col_global, MPIAIJSplit(self, col_global))
            while len(cache) >= 4:                   # a caller handing over a fresh lgmap array per call must not grow device memory
                cache.pop(next(iter(cache)))
            cache[key] = hit
        return hit[1]

    def matplan(self, rowplan, colplan, maps):
        """Cached block-local sparsity for the staged matrix scatter (fd_matplan_create)."""
        self._build()
        key = ("mp", id(maps[0]._base()), id(maps[1]._base()), id(rowplan), id(colplan))
        mp = self._elem_tables.get(key)
        if mp is None:
            mp = MatPlan(self, rowplan, colplan)
            self._elem_tables[key] = mp
        return mp

    def elem_table(self, rmap: Map, cmap: Map, nlayers=0):
        """Device table element (x layer) -> nonzero position in the NODE pattern (fd_csr_elem_offsets)."""
        self._build()
        key = (id(rmap._base()), id(cmap._base()), nlayers)
        t = self._elem_tables.get(key)
        if t is None:
            if self._node_nnz > 2 ** 31 - 1:
                raise _lib.FDHipError("the element -> nonzero table of the direct wrapper holds 32-bit places (pattern of 2^31 entries "
                                      "or more): set configuration['mat_scatter'] = 'search'")
            nent = rmap._base().values_with_halo.shape[0]
            t = DeviceBuffer(nent * max(nlayers, 1) * rmap.arity * cmap.arity * 4)
            ro = co = None
            if nlayers:
                ro = np.asarray(rmap.offset, dtype=np.int32)
                co = np.asarray(cmap.offset, dtype=np.int32)
            _lib.call("fd_csr_elem_offsets", self._node_rowptr.ptr, self._node_colidx.ptr, rmap._base()._dev_values(),
                      cmap._base()._dev_values(), nent, rmap.arity, cmap.arity, nlayers,
                      None if ro is None else ro.ctypes.data, None if co is None else co.ctypes.data, t.ptr, None)
            self._elem_tables[key] = t
        return t


class OcrPlan:
    """Owner-computes-rows plan (fd_ocrplan_*): row-node blocks, their entity instances, the per-instance
    copies of the staged maps with their node plans, and the per-entity row-offset table."""

    def __init__(self, sparsity, rmap: Map, cmap: Map, staged_maps, start, end, row_blocks, lane_threads=0, row_order=None):
        """``row_order``: optional ``RowOrder`` (backend-derived row positions); ``row_blocks`` are then ranges of row
        positions and the wrapper flushes row by row ("ocrp")."""
        self.row_blocks = rb = np.ascontiguousarray(row_blocks, dtype=np.int32)
        nb = len(rb) - 1
        h = ctypes.c_void_p()
        self.row_order = row_order
        if row_order is not None:
            _lib.call("fd_ocrplan_create_ordered", rmap._base()._dev_values(), rmap.arity, int(start), int(end), rb.ctypes.data, nb,
                      self._order_code(lane_threads), row_order.pinv.ptr, row_order.npos, row_order.prowptr.ptr, None, ctypes.byref(h))
        else:
            _lib.call("fd_ocrplan_create", rmap._base()._dev_values(), rmap.arity, int(start), int(end), rb.ctypes.data, nb,
                      self._order_code(lane_threads), None, ctypes.byref(h))
        self.h = h.value
        ni, mi = ctypes.c_int64(), ctypes.c_int32()
        _lib.call("fd_ocrplan_info", self.h, ctypes.byref(ni), ctypes.byref(mi))
        self.ninst, self.max_inst, self.nblocks = ni.value, mi.value, nb
        p = [ctypes.c_void_p() for _ in range(4)]
        _lib.call("fd_ocrplan_arrays", self.h, *[ctypes.byref(x) for x in p])
        self.inst_off, inst_off_host, self.inst_ent, self.rblk = (x.value for x in p)
        self.inst_off_host = np.ctypeslib.as_array(ctypes.cast(inst_off_host, ctypes.POINTER(ctypes.c_int32)), shape=(nb + 1,)).copy()
        # geometry of the row blocks
        rp = sparsity._node_rowptr_host()
        self.rows_end = int(rb[-1])
        self.vals_end = int(rp[rb[-1]])              # (positions cover exactly the rows [0, npos): same end either way)
        if row_order is not None:
            self.max_nnz = int(np.diff(row_order.prowptr_host[rb]).max()) if nb else 0
        else:
            self.max_nnz = int(np.diff(rp[rb]).max()) if nb else 0
        self.max_nown = int(np.diff(rb).max()) if nb else 0
        maxlen = sparsity._max_node_rowlen()
        self.kbytes = 1 if maxlen <= 254 else 2
        self._build_tables(sparsity, rmap, cmap, staged_maps)

    def _imap_of(self, m, staged_maps):
        """(per-instance copy of Map ``m``, its key among the staged maps or None)."""
        for key, mm in staged_maps.items():
            if mm._base() is m._base():
                return self._imaps[key], key
        buf = DeviceBuffer(max(self.ninst, 1) * m.arity * 4)
        _lib.call("fd_gather_rows", m._base()._dev_values(), m.arity, self.inst_ent, self.ninst, buf.ptr, None)
        return buf, None

    def _build_tables(self, sparsity, rmap, cmap, staged_maps):
        """Per-instance copies of the staged maps with their node plans over the instance blocks, and the per-INSTANCE
        row-offset table (position of every (i, j) entry inside its CSR row) in instance order: the wrapper streams it
        coalesced next to the local maps instead of gathering 16-byte rows by entity id."""
        self.plans, self._imaps = {}, {}
        for key, m in staged_maps.items():
            imap = DeviceBuffer(max(self.ninst, 1) * m.arity * 4)
            _lib.call("fd_gather_rows", m._base()._dev_values(), m.arity, self.inst_ent, self.ninst, imap.ptr, None)
            self._imaps[key] = imap
            self.plans[key] = Plan(imap.ptr, 0, int(self.ninst), 0, self.inst_off_host, arity=m.arity)
        self.kidx = DeviceBuffer(max(self.ninst, 1) * rmap.arity * cmap.arity * self.kbytes)
        if self.ninst:
            ir, ic = self._imap_of(rmap, staged_maps)[0], self._imap_of(cmap, staged_maps)[0]
            _lib.call("fd_csr_elem_row_offsets", sparsity._node_rowptr.ptr, sparsity._node_colidx.ptr, ir.ptr, ic.ptr,
                      int(self.ninst), rmap.arity, cmap.arity, self.kbytes, self.kidx.ptr, None)

    def records(self, staged_keys, lbits, kbits, diag, words, nr, nc):
        """Bit-packed instance records (fd_ocr_pack_records) for the staged maps ``staged_keys`` in the wrapper's order:
        device buffer of ``words`` 32-bit words per instance, built once per layout."""
        key = (tuple(staged_keys), tuple(lbits), kbits, bool(diag), words)
        cache = self.__dict__.setdefault("_records", {})
        buf = cache.get(key)
        if buf is None:
            n = len(staged_keys)
            buf = DeviceBuffer(max(self.ninst, 1) * words * 4)
            lm = (ctypes.c_void_p * max(n, 1))(*[self.plans[k_].lmap for k_ in staged_keys])
            ar = (ctypes.c_int32 * max(n, 1))(*[self.plans[k_].arity for k_ in staged_keys])
            lb = (ctypes.c_int32 * max(n, 1))(*[int(b) for b in lbits])
            _lib.call("fd_ocr_pack_records", int(self.ninst), n, lm, ar, lb, self.kidx.ptr, self.kbytes, int(nr), int(nc), int(kbits),
                      1 if diag else 0, None, 0, 0, int(words), buf.ptr, None)
            cache[key] = buf
        return buf

    @staticmethod
    def _order_code(lane_threads):
        """fd_ocrplan_create's ``interleave`` argument for configuration["ocr_order"]."""
        order = str(configuration["ocr_order"])
        if order == "stencil":
            return 1
        if order == "natural":
            return 0
        raise ValueError('configuration["ocr_order"] must be "stencil" or "natural"')

    def __del__(self):
        try:
            if self.h:
                _lib.load().fd_ocrplan_free(self.h)
        except Exception:
            pass


class SlicedOcrPlan:
    """Row-sliced owner-computes-rows plan (fd_ocrplan_create_sliced): instances are (entity, local row), grouped by local
    row inside every row block and padded to whole wavefronts.  ``plans`` = node plans of the staged READ maps over the
    (padded) instance blocks; ``tables(rlg, clg)`` = the per-instance accumulator slots / column positions for one pair of
    lgmaps, built on first use and kept for the last few pairs (a BC set is assembled many times)."""

    MAX_TABLE_SETS = 3

    def __init__(self, sparsity, rmap: Map, cmap: Map, staged_maps, start, end, row_blocks, row_order=None, groups=None):
        """``groups``: a partition of the local rows into groups of one or two, [(a, b) | (a, None), ...] -- two rows per instance
        sharing one evaluation of the local kernel (fd_ocrplan_create_paired); None = one row per instance."""
        self.row_blocks = rb = np.ascontiguousarray(row_blocks, dtype=np.int32)
        nb = len(rb) - 1
        self.row_order = row_order
        self._sp, self._rmap, self._cmap = sparsity, rmap._base(), cmap._base()
        self.block = int(sparsity.dsets[0].cdim) * int(sparsity.dsets[1].cdim)     # scalars per (row node, column node) pair
        self.groups = None if groups is None else tuple((int(a), None if b is None else int(b)) for a, b in groups)
        self.rows_per_inst = 1 if groups is None else 2
        h = ctypes.c_void_p()
        pinv_ = row_order.pinv.ptr if row_order is not None else None
        npos_ = row_order.npos if row_order is not None else 0
        if groups is None:
            _lib.call("fd_ocrplan_create_sliced", self._rmap._dev_values(), rmap.arity, int(start), int(end), rb.ctypes.data, nb,
                      pinv_, npos_, int(configuration["ocrs_interleave"]), None, ctypes.byref(h))
        else:
            if self.block != 1:
                raise ValueError("paired instances serve scalar matrices")
            gr = np.array([[a, 255 if b is None else b] for a, b in self.groups], dtype=np.uint8)
            _lib.call("fd_ocrplan_create_paired", self._rmap._dev_values(), rmap.arity, int(start), int(end), rb.ctypes.data, nb,
                      pinv_, npos_, int(configuration["ocrs_interleave"]), len(gr), gr.ctypes.data, None, ctypes.byref(h))
        self.h = h.value
        ni, mi = ctypes.c_int64(), ctypes.c_int32()
        _lib.call("fd_ocrplan_info", self.h, ctypes.byref(ni), ctypes.byref(mi))
        self.ninst, self.max_inst, self.nblocks = ni.value, mi.value, nb
        p = [ctypes.c_void_p() for _ in range(4)]
        _lib.call("fd_ocrplan_arrays", self.h, *[ctypes.byref(x) for x in p])
        self.inst_off, inst_off_host, self.inst_ent, self.rblk = (x.value for x in p)
        self.inst_off_host = np.ctypeslib.as_array(ctypes.cast(inst_off_host, ctypes.POINTER(ctypes.c_int32)), shape=(nb + 1,)).copy()
        cr, va, nr = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_int64()
        _lib.call("fd_ocrplan_sliced_arrays", self.h, ctypes.byref(cr), ctypes.byref(va), ctypes.byref(nr))
        self.chunk_role, self.valid, self.nreal = cr.value, va.value, nr.value
        rp = sparsity._node_rowptr_host()
        self.rows_end = int(rb[-1])
        self.vals_end = int(rp[rb[-1]]) * self.block
        acc = row_order.prowptr_host if row_order is not None else rp
        self.max_nnz = int(np.diff(acc[rb]).max()) if nb else 0
        self.max_nown = int(np.diff(rb).max()) if nb else 0
        maxlen = sparsity._max_node_rowlen()
        self.kbytes = 1 if maxlen <= 254 else 2
        self.plans, self._imaps = {}, {}
        for key, m in staged_maps.items():
            imap = DeviceBuffer(max(self.ninst, 1) * m.arity * 4)
            _lib.call("fd_gather_rows", m._base()._dev_values(), m.arity, self.inst_ent, self.ninst, imap.ptr, None)
            self._imaps[key] = imap
            self.plans[key] = Plan(imap.ptr, 0, int(self.ninst), 0, self.inst_off_host, arity=m.arity)
        self._tables = {}

    def tables(self, rlg, clg, lgmap_ptr, per_dof=False):
        """(slot, column positions, row lengths or None, row masks or None, column masks or None) for the lgmap objects ``rlg`` /
        ``clg`` (None = no masking); ``per_dof``: the lgmaps are indexed by node*bs + component (``unroll``) and the last two are built;
        ``lgmap_ptr(obj)`` gives the device pointer of an lgmap.  Keyed by object identity (lgmaps are immutable, like the
        reference's PETSc LGMaps); the entry keeps the objects alive so that an id cannot be recycled."""
        def ident(o):
            # host arrays: object identity (the entry keeps them alive); device arrays known by pointer (bridge.DeviceMat.
            # set_lgmaps makes a new wrapper per call, like the reference swaps lgmaps per assemble): pointer + caller's token
            return ("dev", o._fd_dev_ptr, getattr(o, "_fd_token", None)) if hasattr(o, "_fd_dev_ptr") else id(o)
        key = (ident(rlg), ident(clg), bool(per_dof))
        t = self._tables.pop(key, None)
        if t is None:
            if self.rows_per_inst > 1 and (per_dof or self.block > 1):
                raise ValueError("paired instances serve scalar matrices with node lgmaps")
            slot = DeviceBuffer(max(self.ninst, 1) * self.rows_per_inst * 2)
            kk = DeviceBuffer(max(self.ninst, 1) * self.rows_per_inst * self._cmap.arity * self.kbytes)
            rlen = DeviceBuffer(max(self.ninst, 1) * 2) if self.block > 1 else None
            rmask = DeviceBuffer(max(self.ninst, 1)) if per_dof else None
            cmask = DeviceBuffer(max(self.ninst, 1) * 8) if per_dof else None
            sp, ro = self._sp, self.row_order
            _lib.call("fd_ocrplan_sliced_tables", self.h, self._rmap._dev_values(), self._cmap._dev_values(), self._cmap.arity,
                      sp._node_rowptr.ptr, sp._node_colidx.ptr, ro.nstart.ptr if ro is not None else sp._node_rowptr.ptr,
                      ro.prowptr.ptr if ro is not None else sp._node_rowptr.ptr,
                      lgmap_ptr(rlg) if rlg is not None else None, lgmap_ptr(clg) if clg is not None else None,
                      self.kbytes, slot.ptr, rlen.ptr if rlen is not None else None, kk.ptr,
                      int(sp.dsets[0].cdim), int(sp.dsets[1].cdim), rmask.ptr if per_dof else None, cmask.ptr if per_dof else None, None)
            t = (slot, kk, rlen, rmask, cmask, rlg, clg, {})
            while len(self._tables) >= self.MAX_TABLE_SETS:
                self._tables.pop(next(iter(self._tables)))
        self._tables[key] = t                       # most recently used last
        return t[:5]

    @staticmethod
    def choose_groups(rmap, start, end, row_blocks, row_order=None):
        """A partition of the local rows into pairs (and one single row when their number is odd) for two-rows-per-instance plans:
        greedy maximum-weight matching on the co-ownership counts (fd_ocrplan_pair_counts: entities whose rows a and b fall into one
        row block).  Deterministic for a given mesh and block cut; ties go to the lower rows."""
        ar = rmap.arity
        if ar > 32:
            return None
        rb = np.ascontiguousarray(row_blocks, dtype=np.int32)
        cnt = np.zeros(ar * ar, dtype=np.int64)
        _lib.call("fd_ocrplan_pair_counts", rmap._base()._dev_values(), ar, int(start), int(end), rb.ctypes.data, len(rb) - 1,
                  row_order.pinv.ptr if row_order is not None else None, row_order.npos if row_order is not None else 0,
                  cnt.ctypes.data, None)
        cnt = cnt.reshape(ar, ar)
        pairs = sorted(((int(cnt[a, b]), -a, -b) for a in range(ar) for b in range(a + 1, ar)), reverse=True)
        free, groups = set(range(ar)), []
        for _, na, nb_ in pairs:
            a, b = -na, -nb_
            if a in free and b in free:
                groups.append((a, b))
                free -= {a, b}
        groups += [(a, None) for a in sorted(free)]
        return tuple(sorted(groups))

    def records(self, rlg, clg, lgmap_ptr, staged_keys, lbits, kbits, sbits, words):
        """Bit-packed instance records for one pair of lgmaps (scalar matrices, node lgmaps): local-map rows of the staged maps,
        the column positions (dropped = all ones) and the accumulator slot (dropped = all ones) in ``words`` 32-bit words per
        instance (fd_ocr_pack_records), cached with the table set they are packed from."""
        self.tables(rlg, clg, lgmap_ptr)
        t = next(reversed(self._tables.values()))
        key = (tuple(staged_keys), tuple(lbits), kbits, sbits, words)
        buf = t[7].get(key)
        if buf is None:
            n = len(staged_keys)
            buf = DeviceBuffer(max(self.ninst, 1) * words * 4)
            lm = (ctypes.c_void_p * max(n, 1))(*[self.plans[k_].lmap for k_ in staged_keys])
            ar = (ctypes.c_int32 * max(n, 1))(*[self.plans[k_].arity for k_ in staged_keys])
            lb = (ctypes.c_int32 * max(n, 1))(*[int(b) for b in lbits])
            if self.rows_per_inst > 1:
                _lib.call("fd_ocr_pack_records_rows", int(self.ninst), n, lm, ar, lb, t[1].ptr, self.kbytes, self.rows_per_inst, self._cmap.arity,
                          int(kbits), t[0].ptr, int(sbits), int(words), buf.ptr, None)
            else:
                _lib.call("fd_ocr_pack_records", int(self.ninst), n, lm, ar, lb, t[1].ptr, self.kbytes, 1, self._cmap.arity, int(kbits), 0,
                          t[0].ptr, int(sbits), 1, int(words), buf.ptr, None)
            t[7][key] = buf
        return buf

    def __del__(self):
        try:
            if self.h:
                _lib.load().fd_ocrplan_free(self.h)
        except Exception:
            pass


class RowOrder:
    """A backend-derived order of the rows [0, npos) of a sparsity: ``plist[p]`` = p-th row, ``pinv[row]`` = its position,
    ``prowptr`` = CSR row starts in that order (device + host copy).  Built from an entity order by the first-touch rule
    (fd_first_touch_order)."""

    def __init__(self, rmap: Map, order: "DeviceBuffer", n, npos, node_rowptr_host, rowptr_dev=None):
        """First-touch row order under the entity order ``order`` (fd_first_touch_order).  ``rowptr_dev``: device pointer of the
        same row starts when the caller holds one (saves an upload)."""
        self.npos = int(npos)
        self.pinv, self.plist = DeviceBuffer(max(npos, 1) * 4), DeviceBuffer(max(npos, 1) * 4)
        rank = DeviceBuffer(max(npos, 1) * 4)
        _lib.call("fd_first_touch_order", rmap._base()._dev_values(), rmap.arity, order.ptr, int(n), self.npos, self.pinv.ptr,
                  self.plist.ptr, rank.ptr, None)
        self.rank_host = rank.download(np.int32, (self.npos,))       # position (in ``order``) of the entity first touching row p
        self._tables(node_rowptr_host, rowptr_dev)

    @classmethod
    def from_plist(cls, plist: "DeviceBuffer", npos, node_rowptr_host, rowptr_dev=None):
        """A row order given as a device permutation of [0, npos) (a k-d partition of the rows' own positions)."""
        self = cls.__new__(cls)
        self.npos = int(npos)
        self.plist, self.pinv = plist, DeviceBuffer(max(npos, 1) * 4)
        _lib.call("fd_invert_permutation", plist.ptr, self.npos, self.pinv.ptr, None)
        self.rank_host = None
        self._tables(node_rowptr_host, rowptr_dev)
        return self

    def _tables(self, node_rowptr_host, rowptr_dev=None):
        """prowptr (accumulator starts by position, also kept on the host: the block cuts are made there) and the two lookups the
        wrapper needs, flattened so that neither is a dependent chain of loads: nstart[node] = prowptr[pinv[node]] (accumulator
        offset of a row, by NODE), gstart[p] = rowptr[plist[p]] (CSR start, by POSITION) -- fd_row_order_tables, on the device (the
        numpy version cost 0.13 s of every first Jacobian call at 10 M rows)."""
        n1 = max(self.npos, 1)
        nb_ = _lib.NNZ_BYTES
        self.prowptr, self.nstart, self.gstart = DeviceBuffer((self.npos + 1) * nb_), DeviceBuffer(n1 * nb_), DeviceBuffer(n1 * nb_)
        keep = None
        if rowptr_dev is None:
            keep = DeviceBuffer.from_numpy(np.ascontiguousarray(node_rowptr_host, dtype=_lib.NNZ_DTYPE))
            rowptr_dev = keep.ptr
        _lib.call("fd_row_order_tables", self.npos, self.plist.ptr, rowptr_dev, self.prowptr.ptr, self.nstart.ptr, self.gstart.ptr, None)
        self.prowptr_host = self.prowptr.download(_lib.NNZ_DTYPE, (self.npos + 1,))

    def gpos(self):
        """int32 per accumulator entry (rows in position order, entries in CSR order inside a row): its place in the CSR value
        array -- what the whole-entity "ocrp" flush streams (built on first use)."""
        if getattr(self, "_gpos", None) is None:
            if int(self.prowptr_host[-1]) > 2 ** 31 - 1:
                raise _lib.FDHipError("the whole-entity row flush holds 32-bit places: patterns of 2^31 entries and more take the "
                                      "row-sliced shapes (run-coded flush)")
            self._gpos = DeviceBuffer(max(int(self.prowptr_host[-1]), 1) * 4)
            _lib.call("fd_row_entry_positions", self.npos, self.prowptr.ptr, self.gstart.ptr, self._gpos.ptr, None)
        return self._gpos

    def gpos_masked(self, sparsity, clg, lgmap_ptr):
        """``gpos()`` with the column lgmap ``clg`` folded in (fd_row_entry_positions_masked: -2 - place for entries in masked
        columns), kept for the last few lgmaps by identity like the tables of a row-sliced plan."""
        if int(self.prowptr_host[-1]) > 2 ** 31 - 1:
            raise _lib.FDHipError("the whole-entity row flush holds 32-bit places")
        key = ("dev", clg._fd_dev_ptr, getattr(clg, "_fd_token", None)) if hasattr(clg, "_fd_dev_ptr") else id(clg)
        cache = self.__dict__.setdefault("_gpos_masked", {})
        hit = cache.pop(key, None)
        if hit is None:
            buf = DeviceBuffer(max(int(self.prowptr_host[-1]), 1) * 4)
            _lib.call("fd_row_entry_positions_masked", self.npos, self.prowptr.ptr, self.gstart.ptr, sparsity._node_colidx.ptr,
                      lgmap_ptr(clg), buf.ptr, None)
            hit = (buf, clg)                          # (the entry keeps the lgmap alive: its id cannot be recycled)
            while len(cache) >= 3:
                cache.pop(next(iter(cache)))
        cache[key] = hit
        return hit[0]

    def runs(self, row_blocks):
        """Run-coded places of the accumulator entries for the row blocks ``row_blocks`` (host array of nblocks + 1 positions;
        fd_ocr_row_runs): (grun, brun, rdelta device buffers, most runs in one block), built once per set of blocks."""
        rb = np.ascontiguousarray(row_blocks, dtype=np.int32)
        key = rb.tobytes()
        hit = getattr(self, "_runs", None)
        if hit is None or hit[0] != key:
            nb = len(rb) - 1
            grun = DeviceBuffer(max(int(self.prowptr_host[-1]), 1))
            brun = DeviceBuffer((nb + 1) * 4)
            rdelta = DeviceBuffer(max(self.npos, 1) * _lib.NNZ_BYTES)
            rblk = DeviceBuffer.from_numpy(rb)
            nruns, mx = ctypes.c_int32(), ctypes.c_int32()
            _lib.call("fd_ocr_row_runs", self.npos, self.prowptr.ptr, self.gstart.ptr, rblk.ptr, nb, grun.ptr, brun.ptr, rdelta.ptr,
                      ctypes.byref(nruns), ctypes.byref(mx), None)
            hit = self._runs = (key, grun, brun, rdelta, int(mx.value), int(nruns.value))
        return hit[1:]

    def tile_cuts(self, entity_blocks, cap):
        """Row-block boundaries (row positions) at the changes of the entity tile that first touches a row: the rows of one
        block are the rows one box of entities meets first -- a box of rows.  Blocks above ``cap`` accumulator entries are
        halved; rows no entity touches (sorted last) are cut into ``cap``-sized ranges.  None without tile boundaries."""
        if entity_blocks is None or len(entity_blocks) < 2 or self.npos == 0 or self.rank_host is None:
            return None
        r = self.rank_host
        ntouched = int(np.searchsorted(r < 0, True)) if (r < 0).any() else self.npos     # ranks ascend, untouched (-1) last
        tile = np.searchsorted(np.asarray(entity_blocks, dtype=np.int64), r[:ntouched], side="right")
        cuts = np.nonzero(np.diff(tile))[0] + 1
        rb = np.concatenate([[0], cuts, [ntouched]]).astype(np.int64)
        prp = self.prowptr_host.astype(np.int64)
        if ntouched < self.npos:
            tail = np.searchsorted(prp, np.arange(prp[ntouched], prp[self.npos] + cap, cap), side="left")
            rb = np.concatenate([rb, tail[(tail > ntouched) & (tail < self.npos)], [self.npos]])
        rb = np.unique(rb)
        while True:                                   # a tile that owns more than the accumulator budget: halve it
            nn = np.diff(prp[rb])
            big = (nn > cap) & (np.diff(rb) > 1)
            if not big.any():
                return rb
            rb = np.unique(np.concatenate([rb, (rb[:-1] + np.diff(rb) // 2)[big]]))


class MPIAIJSplit:
    """Device arrays of MatCreateMPIAIJWithSplitArrays(comm, m, n, M, N, i, j, a, oi, oj, oa) for the owned rows of a Sparsity
    (pyop2/types/mat.py:741-804 creates the PETSc matrix these would be handed to; firedrake/preconditioners/offload.py:25-131
    moves such a matrix to the device)."""

    def __init__(self, sp, col_global):
        rbs, cbs = sp._dsets[0].cdim, sp._dsets[1].cdim
        self.nrows = sp._dsets[0].set.size * rbs
        self.ncols_owned = sp._dsets[1].set.size * cbs
        cg = None
        if col_global is not None:
            cg = DeviceBuffer.from_numpy(np.ascontiguousarray(col_global, dtype=np.int32))
        VP = ctypes.c_void_p
        drp, dci, orp, oci, ork = VP(), VP(), VP(), VP(), VP()
        dn, on = ctypes.c_int64(), ctypes.c_int64()
        _lib.call("fd_csr_split_mpiaij", self.nrows, sp._rowptr.ptr, sp._colidx.ptr, self.ncols_owned, cg.ptr if cg else None,
                  ctypes.byref(drp), ctypes.byref(dci), ctypes.byref(dn), ctypes.byref(orp), ctypes.byref(oci), ctypes.byref(ork),
                  ctypes.byref(on), None)
        self.d_nnz, self.o_nnz = dn.value, on.value
        self.d_rowptr = DeviceBuffer.wrap(drp.value, (self.nrows + 1) * 4)
        self.d_colidx = DeviceBuffer.wrap(dci.value, max(self.d_nnz, 1) * 4)
        self.o_rowptr = DeviceBuffer.wrap(orp.value, (self.nrows + 1) * 4)
        self.o_colidx = DeviceBuffer.wrap(oci.value, max(self.o_nnz, 1) * 4)
        self.o_rank = DeviceBuffer.wrap(ork.value, max(self.o_nnz, 1) * 4)     # place of every suffix entry in its SORTED off-diagonal row
        self.d_vals = self.o_vals = None
        self._nnz_host = self._onnz_host = None

    def values(self, mat):
        """(diagonal-block values, off-diagonal-block values) of ``mat`` as assembled now (fd_csr_split_values)."""
        if self.d_vals is None:
            self.d_vals, self.o_vals = DeviceBuffer(max(self.d_nnz, 1) * 8), DeviceBuffer(max(self.o_nnz, 1) * 8)
        sp = mat.sparsity
        _lib.call("fd_csr_split_values", self.nrows, sp._rowptr.ptr, mat._values_dev().ptr, self.d_rowptr.ptr, self.o_rowptr.ptr,
                  self.o_rank.ptr, self.d_vals.ptr, self.o_vals.ptr, None)
        return self.d_vals, self.o_vals

    def row_counts(self):
        """(d_nnz, o_nnz) per owned scalar row on the host, downloaded once."""
        if self._nnz_host is None:
            self._nnz_host = np.diff(self.d_rowptr.download(np.int32, (self.nrows + 1,)))
            self._onnz_host = np.diff(self.o_rowptr.download(np.int32, (self.nrows + 1,)))
        return self._nnz_host, self._onnz_host


class MatPlan:
    """Python handle on an fd_matplan_t."""

    def __init__(self, sparsity, rowplan, colplan):
        h = ctypes.c_void_p()
        _lib.call("fd_matplan_create", rowplan.h, colplan.h, sparsity._node_rowptr.ptr, sparsity._node_colidx.ptr,
                  sparsity._node_nnz, None, ctypes.byref(h))
        self.h = h.value
        a, b, c, t = ctypes.c_int32(), ctypes.c_int32(), ctypes.c_int32(), ctypes.c_int64()
        _lib.call("fd_matplan_info", self.h, ctypes.byref(a), ctypes.byref(b), ctypes.byref(c), ctypes.byref(t))
        self.max_nnz, self.max_rowlen, self.kbytes, self.total = a.value, b.value, c.value, t.value
        p = [ctypes.c_void_p() for _ in range(4)]
        _lib.call("fd_matplan_arrays", self.h, *[ctypes.byref(x) for x in p])
        self.mb_off, self.gpos, self.lrp, self.kidx = (x.value for x in p)
        z, nz, nx = ctypes.c_void_p(), ctypes.c_int64(), ctypes.c_int64()
        _lib.call("fd_matplan_zero_list", self.h, ctypes.byref(z), ctypes.byref(nz), ctypes.byref(nx))
        self.zero_list, self.n_zero, self.n_exclusive = z.value, nz.value, nx.value
        self._plans = (rowplan, colplan)      # keep alive

    def __del__(self):
        try:
            if self.h:
                _lib.load().fd_matplan_free(self.h)
        except Exception:
            pass


class Mat:
    """pyop2/types/mat.py:607-985, as a device-resident scalar CSR ("aij")."""

    # non-nested: "one block, itself" (mat.py:87-99, 741-768) WITHOUT storing the list -- ``self._blocks = [[self]]`` is a reference cycle,
    # and a Sparsity / Mat in a cycle keeps its device arrays (and, through its maps, their plans and derived orders) until the cycle
    # collector runs instead of until the last reference goes (tests/test_gpu_leaks.py)
    _nested_blocks = None

    @property
    def _blocks(self):
        return self._nested_blocks if self._nested_blocks is not None else [[self]]

    @_blocks.setter
    def _blocks(self, rows):
        self._nested_blocks = None if (len(rows) == 1 and len(rows[0]) == 1 and rows[0][0] is self) else rows

    def __init__(self, sparsity, dtype=None, name=None):
        if not isinstance(sparsity, Sparsity):
            raise SparsityTypeError("Mat needs a Sparsity")
        _check_name(name)
        self._sparsity = sparsity
        self._dtype = np.dtype(dtype) if dtype is not None else ScalarType
        if self._dtype != ScalarType:
            raise DataTypeError("only float64 matrices are supported (ScalarType)")
        self.name = name or f"mat_#x{id(self):x}"
        self._vals = None
        self._zero_pending = False
        self.dat_version = 0
        self._diag_lists, self._diag_pos = {}, {}      # set_local_diagonal_entries: row lists / diagonal places kept on the device
        # mixed spaces: one Mat per block of the nested sparsity (MATNEST, mat.py:741-768)
        self._blocks = ([[Mat(sparsity[i, j], dtype, f"{self.name}_{i}_{j}") for j in range(sparsity.shape[1])]
                         for i in range(sparsity.shape[0])] if sparsity.nested else [[self]])

    sparsity = property(lambda self: self._sparsity)
    dtype = property(lambda self: self._dtype)

    def __getitem__(self, idx):        # mat.py:663-672
        try:
            i, j = idx
            return self._blocks[i][j]
        except TypeError:
            return self._blocks[idx]

    def __iter__(self):                # blocks in row-major order (mat.py:674-677)
        for row in self._blocks:
            yield from row

    def __str__(self):
        return "OP2 Mat: %s, sparsity (%s), datatype %s" % (self.name, self._sparsity, self._dtype.name)

    def __repr__(self):
        return "Mat(%r, %r, %r)" % (self._sparsity, self._dtype, self.name)

    @property
    def dims(self):
        return self._sparsity.dims

    @property
    def nrows(self):
        return self._sparsity.nrows

    @property
    def ncols(self):
        return self._sparsity.ncols

    @property
    def nblock_rows(self):
        return self._sparsity.dsets[0].set.size

    @property
    def nblock_cols(self):
        return self._sparsity.dsets[1].set.size

    def _values_dev(self):
        """Device values; performs a pending zero() first (anything but a fused staged assembly sees
        the zeroed matrix)."""
        v = self._values_raw()
        if self._zero_pending:
            v.zero()
            self._zero_pending = False
        return v

    def _values_raw(self):
        if self._vals is None:
            self._sparsity._build()
            self._vals = DeviceBuffer(max(self._sparsity._nnz, 1) * 8)
            self._vals.zero()       # fill_with_zeros, sparsity.pyx:162-389
        return self._vals

    def zero(self):                 # mat.py:851-855
        if self._sparsity.nested:
            for blk in self:
                blk.zero()
            return
        self._zero()

    def _zero(self):
        """Zero the matrix.  The memset is deferred: a staged assembly that follows overwrites the entries
        each block owns exclusively and clears only the shared/untouched ones (fd_matplan_zero_list), which
        fuses the zeroing pass into the assembly kernel.  Any other access flushes it."""
        self._values_raw()
        self._zero_pending = True
        self.dat_version += 1

    def assemble(self):             # mat.py:940-954: nothing is stashed off-process here
        _lib.call("fd_device_sync")

    def __call__(self, access, path, lgmaps=None, unroll_map=False):
        from .parloop import MatLegacyArg, MixedMatLegacyArg
        if access not in (WRITE, INC):
            raise ModeValueError("Mat arguments must have access mode WRITE or INC")
        rmap, cmap = path
        if self._sparsity.nested:
            return MixedMatLegacyArg(self, (rmap, cmap), access, lgmaps, bool(unroll_map))
        if configuration["type_check"]:
            if rmap.toset != self._sparsity.dsets[0].set or cmap.toset != self._sparsity.dsets[1].set:
                raise MapValueError("Path maps do not match the Mat's DataSets")
        return MatLegacyArg(self, (rmap, cmap), access, lgmaps, bool(unroll_map))

    # -- host views
    def csr(self):
        """(rowptr, colidx, values) numpy copies."""
        sp = self._sparsity
        vals = self._values_dev().download(np.float64, (sp._nnz,))
        return sp.rowptr, sp.colidx, vals

    def toscipy(self):
        import scipy.sparse as ssp
        rp, ci, v = self.csr()
        return ssp.csr_matrix((v, ci, rp), shape=(self.nrows, self.ncols))

    @property
    def values(self):
        """Dense copy of the owned block (mat.py:968-985)."""
        nr = self.nblock_rows * self._sparsity.dsets[0].cdim
        nc = self.nblock_cols * self._sparsity.dsets[1].cdim
        return np.asarray(self.toscipy().todense())[:nr, :nc]

    def _rows_dev(self, rows):
        if isinstance(rows, Subset):
            rows = rows.indices
        return np.ascontiguousarray(np.asarray(rows, dtype=np.int32).reshape(-1))

    def set_local_diagonal_entries(self, rows, diag_val=1.0, idx=None):   # mat.py:896-937
        r = self._rows_dev(rows)
        rbs = self._sparsity.dsets[0].cdim
        if rbs > 1:
            comps = range(rbs) if idx is None else [idx]
            r = np.concatenate([r * rbs + c for c in comps]).astype(np.int32)
        # a Newton / time loop applies the same row list after every assembly: the list and the places of its diagonal entries
        # stay on the device, keyed by the list's content (a handful of lists per matrix: one per set of boundary conditions)
        key = (len(r), hash(r.tobytes()))
        hit = self._diag_lists.get(key)
        if hit is None:
            if len(self._diag_lists) >= 8:
                self._diag_lists.pop(next(iter(self._diag_lists)))
            hit = self._diag_lists[key] = DeviceBuffer.from_numpy(r)
        self.set_diagonal_rows(hit, len(r), diag_val)
        _lib.call("fd_device_sync")

    def set_diagonal_rows(self, rows_dev, n, diag_val=1.0):
        """``set_local_diagonal_entries`` for a row list that already lives on the device (int32 scalar rows, negative = skip): the
        places of the diagonal entries are searched once per (list, pattern) -- ``fd_csr_diag_positions`` -- and every later call is
        one store per row through them, stream-ordered, no host synchronisation."""
        if n <= 0:
            return
        sp = self._sparsity
        sp._build()
        key = (rows_dev.ptr, int(n))
        pos = self._diag_pos.get(key)
        if pos is None:
            if len(self._diag_pos) >= 8:
                self._diag_pos.pop(next(iter(self._diag_pos)))
            buf = DeviceBuffer(int(n) * 8)
            _lib.call("fd_csr_diag_positions", sp._rowptr.ptr, sp._colidx.ptr, rows_dev.ptr, int(n), buf.ptr, None)
            pos = self._diag_pos[key] = (buf, rows_dev)          # (the list is kept alive with its places: the key is its address)
        _lib.call("fd_csr_set_at", self._values_dev().ptr, pos[0].ptr, int(n), float(diag_val), None)
        self.dat_version += 1

    def zero_rows(self, rows, diag_val=1.0):                               # mat.py:857-891
        r = self._rows_dev(rows)
        rbs = self._sparsity.dsets[0].cdim
        if rbs > 1:
            r = np.concatenate([r * rbs + c for c in range(rbs)]).astype(np.int32)
        sp = self._sparsity
        sp._build()
        d = DeviceBuffer.from_numpy(r)
        _lib.call("fd_csr_zero_rows", sp._rowptr.ptr, sp._colidx.ptr, self._values_dev().ptr, d.ptr, len(r),
                  float(diag_val), None)
        _lib.call("fd_device_sync")
        self.dat_version += 1

    def mpiaij_values(self, col_global=None):
        """The assembled values laid out for MatCreateMPIAIJWithSplitArrays: (split, diagonal-block values, off-diagonal-block
        values), all on the device (Sparsity.mpiaij_split)."""
        split = self._sparsity.mpiaij_split(col_global)
        d, o = split.values(self)
        return split, d, o

    def get_diagonal(self, d: Dat):
        """d = diag(A) on the device (MatGetDiagonal)."""
        sp = self._sparsity
        sp._build()
        _lib.call("fd_csr_get_diagonal", self.nrows, sp._rowptr.ptr, sp._colidx.ptr, self._values_dev().ptr, d._dev_ptr(True), None)

    def mult(self, x: Dat, y: Dat):
        """y = A x on the device (used for the A*x == action(a,x) identity)."""
        sp = self._sparsity
        sp._build()
        _lib.call("fd_csr_spmv", self.nrows, sp._rowptr.ptr, sp._colidx.ptr, self._values_dev().ptr,
                  x._dev_ptr(False), y._dev_ptr(True), None)

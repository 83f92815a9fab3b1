// _pimemb_marshal -- a CPython helper that turns the per-table LISTS of torch tensors an `apply_emb` loop hands over
// (dlrm_s_pytorch.py::apply_emb calls emb_l[k](indices_k, offsets_k) per table; README.md:6,10,14 of the reference)
// into the C ABI's emb_lookup_desc records in one call.  Plumbing only: it computes nothing, owns nothing, and is NOT
// part of libpimemb.so, whose boundary stays free of torch types (include/pimemb.h).  Without it engine.py does the same
// unpacking in Python (~1 us per table and call: data_ptr() alone costs ~0.25 us per tensor).
//
//   pack(desc_addr, table_ids, indices, offsets, outs | None, tables) -> None | (itype, device, common_dim, nbs, key, stream)
//       writes len(table_ids) records at desc_addr (the caller's buffer of 48-byte emb_lookup_desc structs).  Returns
//       None when the arguments are not what this fast path handles (the general Python path then reports the error):
//       every index / offset tensor must be a 1-D contiguous CUDA tensor on one device, int64 or int32 throughout,
//       every output a contiguous float32 CUDA tensor of n_bags x dim elements.  tables: the engine's {id: (rows, dim,
//       dtype)} dict (KeyError for a table that is not loaded).  nbs: bag counts per table (only when outs is None: the
//       caller allocates); key: the records as bytes (only when outs were given) -- an exact call signature for the plan
//       cache; stream: torch's current stream on that device (hipStream_t as an int).
//   fill_pooled(desc_addr, n, base_ptr, dim) -> key
//       pooled pointers of n records laid out back to back from base_ptr (one allocation, one view per table).
//   pack_shard(input_addr, indices, offsets | None, outs, fixed_pooling, dim) -> None | (n_bags, device, stream, itype)
//       the same for the SHARDED call: len(indices) emb_shard_input records (40 bytes each) at input_addr.  1-D contiguous CUDA
//       tensors, int32 (uint32 bits) or int64 throughout (DLRM's dtype: handed over IN PLACE, emb_shard_input.index_type);
//       offsets None = fixed_pooling indices per bag; every table the same number of bags; outs float32 contiguous of
//       n_bags x dim.  ShardedEmbeddingBags.prepare spends ~1.2 us per table in Python without it (32 us for 26 tables against a
//       60-us step).
#define PY_SSIZE_T_CLEAN
#include <Python.h>

#include <cstdint>
#include <cstring>

#include <ATen/core/Tensor.h>
#include <c10/hip/HIPStream.h>
#include <torch/csrc/autograd/python_variable.h>

namespace {

struct Desc {              // emb_lookup_desc (include/pimemb.h), 48 bytes
    uint32_t table_id, fixed_pooling;
    uint64_t indices, offsets, n_indices, n_bags, pooled;
};
static_assert(sizeof(Desc) == 48, "emb_lookup_desc layout");

inline const at::Tensor *as_tensor(PyObject *o) { return THPVariable_Check(o) ? &THPVariable_Unpack(o) : nullptr; }

PyObject *pack(PyObject *, PyObject *args) {
    unsigned long long addr = 0;
    PyObject *ids, *indices, *offsets, *outs, *tables;
    if (!PyArg_ParseTuple(args, "KOOOOO", &addr, &ids, &indices, &offsets, &outs, &tables)) return nullptr;
    if (!PyList_Check(ids) || !PyList_Check(indices) || !PyList_Check(offsets) || !PyDict_Check(tables)) Py_RETURN_NONE;
    const Py_ssize_t n = PyList_GET_SIZE(ids);
    if (n == 0 || PyList_GET_SIZE(indices) != n || PyList_GET_SIZE(offsets) != n) Py_RETURN_NONE;
    const bool have_outs = outs != Py_None;
    if (have_outs && (!(PyList_Check(outs) || PyTuple_Check(outs)) || PySequence_Fast_GET_SIZE(outs) != n)) Py_RETURN_NONE;
    Desc *d = reinterpret_cast<Desc *>(static_cast<uintptr_t>(addr));
    c10::ScalarType st = c10::ScalarType::Undefined;
    c10::DeviceIndex dev = -1;
    long common_dim = -1;
    for (Py_ssize_t k = 0; k < n; k++) {
        const at::Tensor *ia = as_tensor(PyList_GET_ITEM(indices, k)), *oa = as_tensor(PyList_GET_ITEM(offsets, k));
        if (!ia || !oa) Py_RETURN_NONE;
        if (k == 0) {
            st = ia->scalar_type();
            if (st != c10::ScalarType::Long && st != c10::ScalarType::Int) Py_RETURN_NONE;
            if (!ia->is_cuda()) Py_RETURN_NONE;
            dev = ia->device().index();
        }
        if (ia->scalar_type() != st || oa->scalar_type() != st || !ia->is_cuda() || !oa->is_cuda() ||
            ia->device().index() != dev || oa->device().index() != dev || ia->dim() != 1 || oa->dim() != 1 ||
            !ia->is_contiguous() || !oa->is_contiguous())
            Py_RETURN_NONE;
        PyObject *id = PyList_GET_ITEM(ids, k);
        PyObject *info = PyDict_GetItemWithError(tables, id);       // borrowed; (rows, dim, dtype)
        if (!info) {
            if (!PyErr_Occurred()) PyErr_SetObject(PyExc_KeyError, id);
            return nullptr;
        }
        const long table_id = PyLong_AsLong(id);
        const long dim = PyLong_AsLong(PyTuple_GET_ITEM(info, 1));
        if ((table_id == -1 || dim == -1) && PyErr_Occurred()) return nullptr;
        common_dim = (k == 0 || common_dim == dim) ? dim : 0;
        d[k].table_id = (uint32_t)table_id;
        d[k].fixed_pooling = 0;
        d[k].indices = (uint64_t)(uintptr_t)ia->const_data_ptr();
        d[k].offsets = (uint64_t)(uintptr_t)oa->const_data_ptr();
        d[k].n_indices = (uint64_t)ia->numel();
        d[k].n_bags = (uint64_t)oa->numel();
        d[k].pooled = 0;
        if (have_outs) {
            const at::Tensor *ua = as_tensor(PySequence_Fast_GET_ITEM(outs, k));
            if (!ua || ua->scalar_type() != c10::ScalarType::Float || !ua->is_cuda() || ua->device().index() != dev ||
                !ua->is_contiguous() || (uint64_t)ua->numel() != d[k].n_bags * (uint64_t)dim)
                Py_RETURN_NONE;
            d[k].pooled = (uint64_t)(uintptr_t)ua->const_data_ptr();
        }
    }
    PyObject *nbs = Py_None, *key = Py_None;
    if (have_outs) {
        key = PyBytes_FromStringAndSize(reinterpret_cast<const char *>(d), n * (Py_ssize_t)sizeof(Desc));
        Py_INCREF(nbs);
    } else {
        nbs = PyTuple_New(n);
        for (Py_ssize_t k = 0; k < n; k++) PyTuple_SET_ITEM(nbs, k, PyLong_FromUnsignedLongLong(d[k].n_bags));
        Py_INCREF(key);
    }
    const unsigned long long stream = (unsigned long long)(uintptr_t)c10::hip::getCurrentHIPStream(dev).stream();
    return Py_BuildValue("(iilNNK)", st == c10::ScalarType::Long ? 1 : 0, (int)dev, common_dim, nbs, key, stream);
}

PyObject *fill_pooled(PyObject *, PyObject *args) {
    unsigned long long addr = 0, base = 0;
    Py_ssize_t n = 0;
    long dim = 0;
    if (!PyArg_ParseTuple(args, "KnKl", &addr, &n, &base, &dim)) return nullptr;
    Desc *d = reinterpret_cast<Desc *>(static_cast<uintptr_t>(addr));
    uint64_t at = base;
    for (Py_ssize_t k = 0; k < n; k++) {
        d[k].pooled = at;
        at += d[k].n_bags * (uint64_t)dim * 4u;
    }
    return PyBytes_FromStringAndSize(reinterpret_cast<const char *>(d), n * (Py_ssize_t)sizeof(Desc));
}

struct ShardInput {        // emb_shard_input (include/pimemb.h), 40 bytes
    uint64_t indices, offsets, n_indices;
    uint32_t fixed_pooling, index_type;
    uint64_t pooled;
};
static_assert(sizeof(ShardInput) == 40, "emb_shard_input layout");

PyObject *pack_shard(PyObject *, PyObject *args) {
    unsigned long long addr = 0;
    PyObject *indices, *offsets, *outs;
    long fixed_pooling = 0, dim = 0;
    if (!PyArg_ParseTuple(args, "KOOOll", &addr, &indices, &offsets, &outs, &fixed_pooling, &dim)) return nullptr;
    if (!PyList_Check(indices) || !(PyList_Check(outs) || PyTuple_Check(outs)) || dim <= 0) Py_RETURN_NONE;
    const Py_ssize_t n = PyList_GET_SIZE(indices);
    const bool have_off = offsets != Py_None;
    if (n == 0 || PySequence_Fast_GET_SIZE(outs) != n || (have_off && (!PyList_Check(offsets) || PyList_GET_SIZE(offsets) != n))) Py_RETURN_NONE;
    if (!have_off && fixed_pooling <= 0) Py_RETURN_NONE;
    ShardInput *d = reinterpret_cast<ShardInput *>(static_cast<uintptr_t>(addr));
    c10::DeviceIndex dev = -1;
    c10::ScalarType st = c10::ScalarType::Undefined;
    uint64_t n_bags = 0;
    for (Py_ssize_t k = 0; k < n; k++) {
        const at::Tensor *ia = as_tensor(PyList_GET_ITEM(indices, k));
        const at::Tensor *oa = have_off ? as_tensor(PyList_GET_ITEM(offsets, k)) : nullptr;
        const at::Tensor *ua = as_tensor(PySequence_Fast_GET_ITEM(outs, k));
        if (!ia || !ua || (have_off && !oa)) Py_RETURN_NONE;
        if (k == 0) {
            st = ia->scalar_type();
            if (st != c10::ScalarType::Int && st != c10::ScalarType::Long) Py_RETURN_NONE;
            if (!ia->is_cuda()) Py_RETURN_NONE;
            dev = ia->device().index();
        }
        if (ia->scalar_type() != st || !ia->is_cuda() || ia->dim() != 1 || !ia->is_contiguous()) Py_RETURN_NONE;
        if (ia->device().index() != dev) Py_RETURN_NONE;
        uint64_t nb;
        if (have_off) {
            if (oa->scalar_type() != st || !oa->is_cuda() || oa->device().index() != dev || oa->dim() != 1 || !oa->is_contiguous())
                Py_RETURN_NONE;
            nb = (uint64_t)oa->numel();
        } else {
            if ((uint64_t)ia->numel() % (uint64_t)fixed_pooling) Py_RETURN_NONE;
            nb = (uint64_t)ia->numel() / (uint64_t)fixed_pooling;
        }
        if (k == 0) n_bags = nb;
        if (nb != n_bags) Py_RETURN_NONE;                   // (the Python path words the error)
        if (ua->scalar_type() != c10::ScalarType::Float || !ua->is_cuda() || ua->device().index() != dev || !ua->is_contiguous() ||
            (uint64_t)ua->numel() != n_bags * (uint64_t)dim)
            Py_RETURN_NONE;
        d[k].indices = (uint64_t)(uintptr_t)ia->const_data_ptr();
        d[k].offsets = have_off ? (uint64_t)(uintptr_t)oa->const_data_ptr() : 0;
        d[k].n_indices = (uint64_t)ia->numel();
        d[k].fixed_pooling = have_off ? 0u : (uint32_t)fixed_pooling;
        d[k].index_type = st == c10::ScalarType::Long ? 1u : 0u;      // EMB_IDX_I64 : EMB_IDX_U32
        d[k].pooled = (uint64_t)(uintptr_t)ua->const_data_ptr();
    }
    const unsigned long long stream = (unsigned long long)(uintptr_t)c10::hip::getCurrentHIPStream(dev).stream();
    return Py_BuildValue("(KiKi)", (unsigned long long)n_bags, (int)dev, stream, st == c10::ScalarType::Long ? 1 : 0);
}

PyMethodDef methods[] = {
    {"pack", pack, METH_VARARGS, "lists of torch tensors -> emb_lookup_desc records (see the file header)"},
    {"fill_pooled", fill_pooled, METH_VARARGS, "pooled pointers laid out back to back from one allocation"},
    {"pack_shard", pack_shard, METH_VARARGS, "lists of torch tensors -> emb_shard_input records (see the file header)"},
    {nullptr, nullptr, 0, nullptr}};

PyModuleDef module = {PyModuleDef_HEAD_INIT, "_pimemb_marshal", "torch tensor lists -> emb_lookup_desc records", -1, methods,
                      nullptr, nullptr, nullptr, nullptr};

}  // namespace

PyMODINIT_FUNC PyInit__pimemb_marshal(void) { return PyModule_Create(&module); }

// CPython extension `_kSpider_internal` — the module name and call signature of the
// reference's SWIG wrapper (/root/reference/src/swig_interfaces/kSpider_internal.i:1,11;
// built with -keyword, setup.py:140-146, so keyword arguments work as in
// test/kspider_run.py:4).  Only pairwise() is implemented; the other ten functions of the
// reference module raise NotImplementedError.
#define PY_SSIZE_T_CLEAN
#include <Python.h>

#include "../../include/kspider_amd.h"

static PyObject* py_pairwise(PyObject*, PyObject* args, PyObject* kwargs) {
    static const char* kw[] = {"index_prefix", "user_threads", nullptr};
    const char* prefix = nullptr;
    int threads = 1;
    if (!PyArg_ParseTupleAndKeywords(args, kwargs, "si:pairwise", const_cast<char**>(kw), &prefix, &threads))
        return nullptr;
    int rc = kspider_pairwise(prefix, threads);   // GIL held, like the SWIG wrapper (no -threads)
    if (rc != KSP_OK) {
        PyErr_SetString(PyExc_RuntimeError, ksp_last_error());
        return nullptr;
    }
    Py_RETURN_NONE;
}

static PyObject* py_unavailable(PyObject*, PyObject*, PyObject*) {
    PyErr_SetString(PyExc_NotImplementedError,
                    "kspider_amd implements kSpider.pairwise() only; indexing/sketching stay with the reference build");
    return nullptr;
}

#define KSP_STUB(name) {name, (PyCFunction)(void (*)(void))py_unavailable, METH_VARARGS | METH_KEYWORDS, "not implemented"}
static PyMethodDef methods[] = {
    {"pairwise", (PyCFunction)(void (*)(void))py_pairwise, METH_VARARGS | METH_KEYWORDS,
     "pairwise(index_prefix, user_threads) -> None"},
    KSP_STUB("index_kmers"), KSP_STUB("index_kmers_nonCanonical"), KSP_STUB("index_skipmers"),
    KSP_STUB("index_protein"), KSP_STUB("index_dayhoff"), KSP_STUB("index_datasets"),
    KSP_STUB("sourmash_sigs_indexing"), KSP_STUB("paired_end_to_kDataFrame"),
    KSP_STUB("single_end_to_kDataFrame"), KSP_STUB("protein_to_kDataFrame"),
    {nullptr, nullptr, 0, nullptr}};

static struct PyModuleDef moddef = {PyModuleDef_HEAD_INIT, "_kSpider_internal",
                                    "MI355X-native kSpider pairwise (kspider_amd)", -1, methods,
                                    nullptr, nullptr, nullptr, nullptr};

PyMODINIT_FUNC PyInit__kSpider_internal(void) { return PyModule_Create(&moddef); }

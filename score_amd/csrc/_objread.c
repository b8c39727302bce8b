/* _objread -- one pass over a list of measurement objects, several attributes per object.
 *
 * The object API of solve_score (FactorGraphData: lists of PoseMeasurement / FGRangeMeasurement objects,
 * /root/reference/score/utils/gurobi_utils.py:380-404, :449-501 iterate them one by one) is turned into the flat arrays
 * of `score_graph` by score_amd/native.py::graph_arrays.  With numpy.fromiter(map(attrgetter(..))) that is one pass per
 * ATTRIBUTE -- 11 passes over 47 k objects on the headline graph, every pass re-walking objects scattered over the heap.
 * gather() walks the list once, reads every requested attribute while the object (and its __dict__) is in cache, and
 * skips the generic attribute protocol for plain instance attributes (dataclass / attrs instances: value straight from
 * the instance dict unless the type defines a data descriptor of that name).
 *
 *   gather(seq, names, kinds, outs, maps)
 *     seq    list or tuple of objects (n of them)
 *     names  tuple of attribute names (str)
 *     kinds  str, one character per name:
 *              'd'  float  -> outs[j]: writable float64 buffer of n
 *              'i'  key    -> outs[j]: writable int32 buffer of n, value = maps[j][attr]      (KeyError names the key)
 *              'p'  pair of keys (any 2-sequence) -> outs[j]: writable int32 buffer of 2 n (row-major (n, 2)),
 *                   values = maps[j][attr[0]], maps[j][attr[1]]
 *              'o'  object -> outs[j]: list of n (slot i receives the attribute; 2-sequences that are not tuples
 *                   are stored as tuples when kinds[j] == 'O')
 *     maps   tuple, dict or None per name
 *
 * Pure host glue (CPython C API); score_amd/native.py falls back to the fromiter passes when it is not built.
 */
#define PY_SSIZE_T_CLEAN
#include <Python.h>
#include <stdint.h>

#define MAXN 16

static PyObject* fetch(PyObject* obj, PyObject* name, PyTypeObject** seen, int* plain) {
    /* plain[.] is decided once per (type, name): no data descriptor of that name on the type */
    PyTypeObject* tp = Py_TYPE(obj);
    if (*seen != tp) {
        PyObject* descr = _PyType_Lookup(tp, name); /* borrowed */
        /* a class with its own __getattribute__ / __getattr__ slot decides for itself: only the generic protocol may be
           short-cut through the instance dict */
        *plain = tp->tp_getattro == PyObject_GenericGetAttr && !(descr && Py_TYPE(descr)->tp_descr_set);
        *seen = tp;
    }
    if (*plain) {
        PyObject** dp = _PyObject_GetDictPtr(obj);
        if (dp && *dp) {
            PyObject* v = PyDict_GetItemWithError(*dp, name); /* borrowed */
            if (v) { Py_INCREF(v); return v; }
            if (PyErr_Occurred()) return NULL;
        }
    }
    return PyObject_GetAttr(obj, name);
}

static int map_key(PyObject* map, PyObject* key, int32_t* out) {
    PyObject* v = PyDict_GetItemWithError(map, key); /* borrowed */
    if (!v) {
        if (!PyErr_Occurred()) PyErr_SetObject(PyExc_KeyError, key);
        return -1;
    }
    const long x = PyLong_AsLong(v);
    if (x == -1 && PyErr_Occurred()) return -1;
    *out = (int32_t)x;
    return 0;
}

static PyObject* gather(PyObject* self, PyObject* args) {
    PyObject *seq, *names, *outs, *maps;
    const char* kinds;
    Py_ssize_t nk = 0;
    (void)self;
    if (!PyArg_ParseTuple(args, "OO!s#O!O!", &seq, &PyTuple_Type, &names, &kinds, &nk, &PyTuple_Type, &outs, &PyTuple_Type, &maps)) return NULL;
    const Py_ssize_t m = PyTuple_GET_SIZE(names);
    if (m > MAXN || nk != m || PyTuple_GET_SIZE(outs) != m || PyTuple_GET_SIZE(maps) != m) {
        PyErr_SetString(PyExc_ValueError, "gather: names, kinds, outs and maps must have the same length (<= 16)");
        return NULL;
    }
    PyObject* fast = PySequence_Fast(seq, "gather: seq must be a sequence");
    if (!fast) return NULL;
    const Py_ssize_t n = PySequence_Fast_GET_SIZE(fast);
    PyObject** items = PySequence_Fast_ITEMS(fast);
    Py_buffer buf[MAXN];
    int have[MAXN] = {0};
    PyTypeObject* seen[MAXN] = {0};
    int plain[MAXN] = {0};
    int ok = 1;
    for (Py_ssize_t j = 0; j < m && ok; ++j) {
        PyObject* o = PyTuple_GET_ITEM(outs, j);
        PyObject* mp = PyTuple_GET_ITEM(maps, j);
        const char k = kinds[j];
        if (!PyUnicode_Check(PyTuple_GET_ITEM(names, j))) { PyErr_SetString(PyExc_TypeError, "gather: names must be str"); ok = 0; break; }
        if (k == 'o' || k == 'O') {
            if (!PyList_Check(o) || PyList_GET_SIZE(o) != n) { PyErr_SetString(PyExc_ValueError, "gather: 'o' needs a list of len(seq)"); ok = 0; }
            continue;
        }
        if (k != 'd' && k != 'i' && k != 'p') { PyErr_SetString(PyExc_ValueError, "gather: unknown kind"); ok = 0; break; }
        if ((k == 'i' || k == 'p') && !PyDict_Check(mp)) { PyErr_SetString(PyExc_TypeError, "gather: 'i' / 'p' need a dict"); ok = 0; break; }
        if (PyObject_GetBuffer(o, &buf[j], PyBUF_WRITABLE | PyBUF_C_CONTIGUOUS | PyBUF_FORMAT) != 0) { ok = 0; break; }
        have[j] = 1;
        const Py_ssize_t want = (k == 'd') ? n * 8 : (k == 'i' ? n * 4 : n * 8);
        if (buf[j].len != want) { PyErr_SetString(PyExc_ValueError, "gather: output buffer has the wrong size"); ok = 0; }
        /* the element type as well as the byte length: a float32 or int64 array of the right byte size would be filled
           with misread data (format: native float64 for 'd', native int32 for 'i' / 'p') */
        const char* f = buf[j].format ? buf[j].format : "B";
        if (*f == '@' || *f == '=' || *f == '<') ++f;
        const int fmt_ok = (k == 'd') ? (buf[j].itemsize == 8 && f[0] == 'd' && !f[1])
                                      : (buf[j].itemsize == 4 && (f[0] == 'i' || f[0] == 'l') && !f[1]);
        if (ok && !fmt_ok) { PyErr_SetString(PyExc_TypeError, "gather: output buffer must be float64 ('d') or int32 ('i', 'p')"); ok = 0; }
    }
    for (Py_ssize_t i = 0; i < n && ok; ++i) {
        PyObject* obj = items[i];
        for (Py_ssize_t j = 0; j < m; ++j) {
            PyObject* v = fetch(obj, PyTuple_GET_ITEM(names, j), &seen[j], &plain[j]);
            if (!v) { ok = 0; break; }
            const char k = kinds[j];
            if (k == 'd') {
                const double x = PyFloat_AsDouble(v);
                if (x == -1.0 && PyErr_Occurred()) ok = 0;
                else ((double*)buf[j].buf)[i] = x;
                Py_DECREF(v);
            } else if (k == 'i') {
                if (map_key(PyTuple_GET_ITEM(maps, j), v, &((int32_t*)buf[j].buf)[i]) != 0) ok = 0;
                Py_DECREF(v);
            } else if (k == 'p') {
                PyObject* pr = PySequence_Fast(v, "gather: 'p' attribute must be a pair");
                Py_DECREF(v);
                if (!pr) { ok = 0; break; }
                if (PySequence_Fast_GET_SIZE(pr) != 2) { PyErr_SetString(PyExc_ValueError, "gather: 'p' attribute must be a pair"); ok = 0; }
                else if (map_key(PyTuple_GET_ITEM(maps, j), PySequence_Fast_GET_ITEM(pr, 0), &((int32_t*)buf[j].buf)[2 * i]) != 0 ||
                         map_key(PyTuple_GET_ITEM(maps, j), PySequence_Fast_GET_ITEM(pr, 1), &((int32_t*)buf[j].buf)[2 * i + 1]) != 0) ok = 0;
                Py_DECREF(pr);
            } else { /* 'o' / 'O' */
                if (k == 'O' && !PyTuple_CheckExact(v)) {
                    PyObject* t = PySequence_Tuple(v);
                    Py_DECREF(v);
                    v = t;
                    if (!v) { ok = 0; break; }
                }
                PyObject* lst = PyTuple_GET_ITEM(outs, j);
                PyObject* old = PyList_GET_ITEM(lst, i);
                PyList_SET_ITEM(lst, i, v); /* steals v */
                Py_XDECREF(old);
            }
            if (!ok) break;
        }
    }
    for (Py_ssize_t j = 0; j < m; ++j)
        if (have[j]) PyBuffer_Release(&buf[j]);
    Py_DECREF(fast);
    if (!ok) return NULL;
    Py_RETURN_NONE;
}

static PyMethodDef methods[] = {
    {"gather", gather, METH_VARARGS, "gather(seq, names, kinds, outs, maps): several attributes of every object of seq in one pass"},
    {NULL, NULL, 0, NULL},
};

static struct PyModuleDef moduledef = {PyModuleDef_HEAD_INIT, "_objread", "one-pass attribute reads over lists of measurement objects", -1, methods, NULL, NULL, NULL, NULL};

PyMODINIT_FUNC PyInit__objread(void) { return PyModule_Create(&moduledef); }

/* _aegis_pyevents: the reference's note-event schema -- a list of dicts per clip
 * (/root/reference/aegis_engine_core/midi_logic.py:96-107: note, start, end, confidence, velocity, track, rms_energy, and the
 * technique / slope keys midi_logic.py:100-107, 133-146 add) -- built from the packed events aegis_extract_events returns
 * (include/aegis_hip.h aegis_event, 48 bytes).  Host-side convenience for events_native.extract_batch: a folder's 92 k events
 * took ~45 ms as a Python comprehension, the time of an eighth of the folder's analysis; here one pass over the records with
 * interned keys.  `confidence` and `rms_energy` arrive as lists of the NumPy scalars the reference's events carry
 * (np.float64 / np.float32: list(array) on the Python side), everything else is read from the records.
 * Built by csrc/Makefile into the package directory; events_native falls back to the comprehension when it is absent. */
#define PY_SSIZE_T_CLEAN
#include <Python.h>
#include <stdint.h>
#include <string.h>

typedef struct {
    int32_t clip, note, start, end, velocity;
    uint8_t track, technique, reserved0, reserved1;
    float rms_energy;
    int32_t reserved2;
    double confidence, slope;
} event_rec;

static PyObject *k_note, *k_start, *k_end, *k_conf, *k_vel, *k_track, *k_rms, *k_tech, *k_slope, *v_main, *v_safe;

static PyObject *event_dicts(PyObject *self, PyObject *args) {
    Py_buffer buf;
    PyObject *conf, *energy, *tech;
    if (!PyArg_ParseTuple(args, "y*O!O!O!", &buf, &PyList_Type, &conf, &PyList_Type, &energy, &PyTuple_Type, &tech)) return NULL;
    PyObject *out = NULL;
    if (buf.len % (Py_ssize_t)sizeof(event_rec) != 0) { PyErr_SetString(PyExc_ValueError, "buffer is not a whole number of 48-byte events"); goto done; }
    const Py_ssize_t n = buf.len / (Py_ssize_t)sizeof(event_rec);
    if (PyList_GET_SIZE(conf) != n || PyList_GET_SIZE(energy) != n) { PyErr_SetString(PyExc_ValueError, "confidence / rms_energy lists do not match the events"); goto done; }
    const Py_ssize_t n_tech = PyTuple_GET_SIZE(tech);
    out = PyList_New(n);
    if (!out) goto done;
    const char *base = (const char *)buf.buf;
    for (Py_ssize_t i = 0; i < n; ++i) {
        event_rec e;
        memcpy(&e, base + i * (Py_ssize_t)sizeof(event_rec), sizeof e);
        if (e.technique >= n_tech) { PyErr_SetString(PyExc_ValueError, "technique code out of range"); Py_CLEAR(out); goto done; }
        PyObject *d = _PyDict_NewPresized(9);
        PyObject *a = PyLong_FromLong(e.note), *b = PyLong_FromLong(e.start), *c = PyLong_FromLong(e.end), *v = PyLong_FromLong(e.velocity);
        PyObject *sl = PyFloat_FromDouble(e.slope);
        int bad = !d || !a || !b || !c || !v || !sl;
        /* insertion order = the order the comprehension in events_native wrote them */
        bad = bad || PyDict_SetItem(d, k_note, a) || PyDict_SetItem(d, k_start, b) || PyDict_SetItem(d, k_end, c) ||
              PyDict_SetItem(d, k_conf, PyList_GET_ITEM(conf, i)) || PyDict_SetItem(d, k_vel, v) ||
              PyDict_SetItem(d, k_track, e.track ? v_main : v_safe) || PyDict_SetItem(d, k_rms, PyList_GET_ITEM(energy, i)) ||
              PyDict_SetItem(d, k_tech, PyTuple_GET_ITEM(tech, e.technique)) || PyDict_SetItem(d, k_slope, sl);
        Py_XDECREF(a); Py_XDECREF(b); Py_XDECREF(c); Py_XDECREF(v); Py_XDECREF(sl);
        if (bad) { Py_XDECREF(d); Py_CLEAR(out); if (!PyErr_Occurred()) PyErr_NoMemory(); goto done; }
        PyList_SET_ITEM(out, i, d);
    }
done:
    PyBuffer_Release(&buf);
    return out;
}

static PyMethodDef methods[] = {
    {"event_dicts", event_dicts, METH_VARARGS, "event_dicts(records bytes-like, confidence list, rms_energy list, technique names tuple) -> list of dicts"},
    {NULL, NULL, 0, NULL}};
static struct PyModuleDef moddef = {PyModuleDef_HEAD_INIT, "_aegis_pyevents", NULL, -1, methods};

PyMODINIT_FUNC PyInit__aegis_pyevents(void) {
    PyObject *m = PyModule_Create(&moddef);
    if (!m) return NULL;
#define K(var, s) if (!(var = PyUnicode_InternFromString(s))) return NULL
    K(k_note, "note"); K(k_start, "start"); K(k_end, "end"); K(k_conf, "confidence"); K(k_vel, "velocity"); K(k_track, "track");
    K(k_rms, "rms_energy"); K(k_tech, "technique"); K(k_slope, "slope"); K(v_main, "main"); K(v_safe, "safe");
#undef K
    return m;
}

"""The list ComputeLineTracks / GetTracks return builds its LineTrack objects when they are first looked at
(limap_amd/triangulation.py::_LazyTrackList): it has to behave like the plain list of the reference's binding
(bindings.cc: std::vector<LineTrack> -> list) whichever way a caller gets at its elements."""
import copy
import pickle

import numpy as np

from limap_amd import triangulation as tri


def _arrays(n=6):
    return {"off": np.arange(n + 1) * 2, "line": np.arange(7.0 * n).reshape(n, 7), "image_ids": np.zeros(2 * n, np.int32),
            "line_ids": np.arange(2 * n, dtype=np.int32), "node_ids": np.arange(2 * n), "scores": np.ones(2 * n),
            "line3d": np.zeros((2 * n, 10))}, {0: np.zeros((2 * n, 4))}


def _ids(seq):
    return [tr.line_id_list[0] // 2 for tr in seq]


def test_lazy_track_list_is_a_list():
    t, segs = _arrays()
    n = 6
    L = tri._LazyTrackList(t, segs)
    assert isinstance(L, list) and len(L) == n
    assert list.__getitem__(L, 2) is None            # nothing built yet
    assert _ids([L[2]]) == [2] and _ids([L[-1]]) == [5] and _ids(L[1:3]) == [1, 2]
    assert L[2] is L[2]                               # built once
    assert _ids(L) == list(range(n)) and _ids(reversed(L)) == list(range(n))[::-1]
    assert _ids(list(tri._LazyTrackList(t, segs))) == list(range(n))
    assert _ids(tuple(tri._LazyTrackList(t, segs))) == list(range(n))
    assert _ids(sorted(tri._LazyTrackList(t, segs), key=lambda tr: -tr.line_id_list[0])) == list(range(n))[::-1]
    for bad in (n, -n - 1):
        try:
            tri._LazyTrackList(t, segs)[bad]
            assert False
        except IndexError:
            pass


def test_lazy_track_list_copies_and_mutations():
    t, segs = _arrays()
    n = 6
    L = tri._LazyTrackList(t, segs)
    first = L[0]
    c = L.copy()
    assert type(c) is tri._LazyTrackList and c[0] is first and _ids([c[3]]) == [3]
    assert list.__getitem__(L, 3) is None              # the copy built its own
    assert _ids(copy.copy(tri._LazyTrackList(t, segs))) == list(range(n))
    assert _ids(copy.deepcopy(tri._LazyTrackList(t, segs))) == list(range(n))
    p = pickle.loads(pickle.dumps(tri._LazyTrackList(t, segs)))
    assert isinstance(p, list) and _ids(p) == list(range(n)) and p[1].line.start[0] == 7.0
    r = tri._LazyTrackList(t, segs); r.reverse(); assert _ids(r) == list(range(n))[::-1]
    q = tri._LazyTrackList(t, segs); assert _ids([q.pop()]) == [5] and len(q) == 5 and _ids(q) == [0, 1, 2, 3, 4]
    a = tri._LazyTrackList(t, segs) + ["x"]; assert a[-1] == "x" and _ids(a[:-1]) == list(range(n))
    m = tri._LazyTrackList(t, segs); assert m[4] in m and m.index(m[4]) == 4 and m.count(m[1]) == 1
    e = tri._LazyTrackList(t, segs); e.append("y"); assert e[-1] == "y" and len(e) == n + 1 and _ids(e[:n]) == list(range(n))
    s = tri._LazyTrackList(t, segs); s.sort(key=lambda tr: -tr.line_id_list[0]); assert _ids(s) == list(range(n))[::-1]
    d = tri._LazyTrackList(t, segs); del d[0]; assert _ids(d) == [1, 2, 3, 4, 5]
    assert repr(tri._LazyTrackList(t, segs)).count("LineTrack") == n

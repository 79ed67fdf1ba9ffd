"""FragmentStreamer's lane bookkeeping without a GPU: a lane number the streamer hands to the runner must
never be the number of a lane that still exists -- two jobs would share one capacity bucket and overwrite each other's
descriptors.  The streamer is built around a fake runner (no pipeline handle, no device)."""
import threading

from imfnet_amd.stream import FragmentStreamer


class _Bucket:
    def __init__(self, key):
        self.key, self.in_flight, self.last_use = key, False, 0


class _Runner:
    """FragmentRunner.bucket / touch / drop_bucket as in imfnet_amd/model/graph.py, on plain objects."""

    def __init__(self):
        self.buckets, self.tick, self.streamers = {}, 0, []

    def bucket(self, key, dev, stream=None, lane=0):
        bk = key if lane == 0 else (key, lane)
        self.tick += 1
        b = self.buckets.get(bk)
        if b is None:
            b = self.buckets[bk] = _Bucket(key)
        b.last_use = self.tick
        return b

    def touch(self, b):
        self.tick += 1
        b.last_use = self.tick

    def drop_bucket(self, bk):
        b = self.buckets.pop(bk, None)
        if b is not None:
            for st in self.streamers:
                st.forget(b)


def _streamer(n_buckets=3):
    st = object.__new__(FragmentStreamer)
    st.runner, st.device, st.n_buckets, st.main, st.handle = _Runner(), None, n_buckets, None, None
    st._free, st._made, st._used, st._tick, st._inflight, st._lock = {}, {}, {}, 0, [], threading.RLock()
    st.runner.streamers.append(st)
    return st


def test_a_dropped_lane_number_is_reissued_not_a_live_one():
    st = _streamer()
    a, b, c = (st._acquire("K") for _ in range(3))
    assert {x.lane for x in (a, b, c)} == {1, 2, 3} and len({id(x) for x in (a, b, c)}) == 3
    for x in (a, b, c):
        st._release(x)
    st.runner.drop_bucket(("K", 2))                     # the runner's own eviction took lane 2 away
    assert st._made["K"] == {1, 3} and all(x.lane != 2 for x in st._free["K"])
    got = [st._acquire("K") for _ in range(3)]          # three jobs in flight again
    assert len({id(x) for x in got}) == 3, "two jobs share one capacity bucket"
    assert sorted(x.lane for x in got) == [1, 2, 3] and all(x.in_flight for x in got)
    assert st.runner.buckets[("K", 2)] is next(x for x in got if x.lane == 2)


def test_lane_one_dropped_while_others_are_in_flight():
    st = _streamer()
    a, b = st._acquire("K"), st._acquire("K")
    st._release(a)                                      # lane 1 idle, lane 2 in flight
    st.runner.drop_bucket(("K", a.lane))
    c = st._acquire("K")
    assert c is not b and c.lane == a.lane == 1 and st._made["K"] == {1, 2}
    d = st._acquire("K")
    assert d.lane == 3 and len({id(b), id(c), id(d)}) == 3


def test_a_lane_in_use_is_the_most_recently_used_for_the_runner():
    st = _streamer()
    a = st._acquire("K")
    st._release(a)
    other = st.runner.bucket("direct", None)            # something else was used since
    assert other.last_use > a.last_use
    again = st._acquire("K")                            # popped from the idle list: must refresh its age
    assert again is a and a.last_use > other.last_use


def test_key_eviction_drops_exactly_the_live_lanes_and_forgets_the_key():
    st = _streamer()
    st.MAX_KEYS = 2
    for key in ("A", "B"):
        x, y = st._acquire(key), st._acquire(key)
        st._release(x)
        st._release(y)
    st.runner.drop_bucket(("A", 1))                     # A keeps lane 2 only
    z = st._acquire("C")                                # a third key: the least recently used idle key (A) goes
    assert "A" not in st._made and "A" not in st._free and ("A", 2) not in st.runner.buckets
    assert set(st._made) == {"B", "C"} and z.lane == 1
    st.fill_lanes()
    assert st._made["B"] == {1, 2, 3} and st._made["C"] == {1, 2, 3} and len(st._free["C"]) == 2

"""Host logic of the arroy-surface mirror that needs no GPU (arroy_amd/index.py)."""
import pytest

from arroy_amd import distances as D
from arroy_amd import index as I


def test_guess_right_number_of_tree_use_specified_number_of_trees():
    """src/tests/writer.rs:14-29."""
    for n in (1, 10, 100):
        assert I.target_n_trees(n, 768, 100, 3) == n


def test_guess_right_number_of_tree_while_growing():
    """src/tests/writer.rs:31-79: the pinned values of the fitted formula."""
    sizes = [1, 10, 100, 1000, 10_000, 100_000, 1_000_000, 10_000_000, 100_000_000]
    expect = {768: [1, 1, 2, 16, 237, 473, 946, 1892, 3784],
              1512: [1, 1, 2, 16, 152, 304, 608, 1215, 2429],
              3072: [1, 1, 2, 16, 180, 360, 720, 1440, 2879]}
    for dim, vals in expect.items():
        assert [I.target_n_trees(None, dim, n, 0) for n in sizes] == vals


def test_do_not_shrink_by_less_than_twenty_percent():
    """src/writer.rs:1383-1390."""
    assert I.target_n_trees(None, 768, 1000, 18) == 18   # 16 wanted, removing 2 of 18 (12.5 %) is not worth it
    assert I.target_n_trees(None, 768, 1000, 30) == 16   # 14/16 > 20 %: shrink


def test_open_unfinished_db():
    """src/tests/reader.rs:31-43."""
    db = I.Database(D.Euclidean)
    w = I.Writer(db, 0, 2)
    w.add_item(0, [0.0, 0.0])
    with pytest.raises(I.MissingMetadata) as e:
        I.Reader.open(db, 0)
    assert str(e.value) == ("Metadata are missing on index 0, You must build your database before attempting to "
                            "read it")
    assert w.need_build() and not w.is_empty() and w.contains_item(0) and not w.contains_item(1)


def test_add_item_checks_dimensions():
    """src/writer.rs:380-386 -> Error::InvalidVecDimension."""
    w = I.Writer(I.Database(D.Cosine), 0, 3)
    with pytest.raises(I.InvalidVecDimension) as e:
        w.add_item(0, [1.0, 2.0])
    assert str(e.value) == "Invalid vector dimensions. Got 2 but expected 3"
    assert (e.value.expected, e.value.received) == (3, 2)
    assert w.del_item(0) is False

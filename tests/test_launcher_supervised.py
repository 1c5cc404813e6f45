"""The supervising launcher of a generation run (saspa_aug_amd/launcher.py: launch_supervised; advisor finding, round 5): a rank
killed by a signal -- the HIP runtime's graph-replay segmentation fault of long sessions -- makes the launcher start ALL ranks
again, once, as fresh child processes with SASPA_FORK=0; an ordinary failure is not retried.  CPU only: the "rank" is a small
script that kills itself with SIGSEGV unless SASPA_FORK=0."""
import os
import sys
import textwrap

import saspa_aug_amd  # noqa: F401
from saspa_aug_amd import launcher


def _script(tmp_path, body):
    p = tmp_path / "rank.py"
    p.write_text(textwrap.dedent(body))
    return str(p)


def test_signal_death_restarts_fresh_children_with_single_branch_graphs(tmp_path, capfd):
    log = tmp_path / "log.txt"
    script = _script(tmp_path, f"""
        import os, signal
        with open({str(log)!r}, "a") as f:
            f.write(f"{{os.environ['RANK']}} {{os.environ['WORLD_SIZE']}} {{os.environ.get('SASPA_FORK', 'unset')}} {{os.getpid()}}\\n")
        if os.environ.get("SASPA_FORK") != "0" and os.environ["RANK"] == "1":
            os.kill(os.getpid(), signal.SIGSEGV)
        if os.environ["RANK"] == "0":
            print("rank0 done")
    """)
    os.environ.pop("SASPA_FORK", None)
    rc = launcher.launch_supervised(2, script, [])
    out, err = capfd.readouterr()
    assert rc == 0
    assert "died on signal 11" in err and "SASPA_FORK=0" in err
    rows = [r.split() for r in log.read_text().splitlines()]
    first = [r for r in rows if r[2] == "unset"]
    second = [r for r in rows if r[2] == "0"]
    assert sorted(r[0] for r in second) == ["0", "1"] and len(first) >= 1          # both ranks restarted, with the fallback
    assert not {r[3] for r in first} & {r[3] for r in second}                     # fresh processes, not re-used ones
    assert "rank0 done" in out


def test_ordinary_failure_is_returned_not_retried(tmp_path, capfd):
    log = tmp_path / "log.txt"
    script = _script(tmp_path, f"""
        import sys
        open({str(log)!r}, "a").write("x\\n")
        sys.exit(3)
    """)
    assert launcher.launch_supervised(1, script, []) == 3
    assert log.read_text().count("x") == 1


def test_second_signal_death_is_final(tmp_path, capfd):
    script = _script(tmp_path, """
        import os, signal
        os.kill(os.getpid(), signal.SIGSEGV)
    """)
    rc = launcher.launch_supervised(1, script, [])
    assert rc == -11

"""Liveness of ONE other robot, judged from its heartbeat topic.

Behavioural twin of cslam/neighbor_monitor.py:4-53 (same attributes, topic name and truth table,
pinned by tests/golden/gdlcd_ref.json); written around an explicit little state machine:

    silent ──first beat──▶ warming up ──init delay over──▶ alive ◀──▶ stale (no beat for max delay)
"""
import time

from .ros_compat import UInt32

HEARTBEAT_TOPIC = "/r{rid}/cslam/heartbeat"
_QUEUE_DEPTH = 10


class NeighborMonitor(object):
    # bookkeeping of the publication path (NeighborManager reads and advances these)
    last_keyframe_received = -1
    last_keyframe_sent = -1
    last_match_sent = -1

    def __init__(self, node, rid, is_enabled, init_delay_sec, max_delay_sec):
        now = time.time()
        self.node, self.robot_id = node, rid
        self.is_enabled = is_enabled
        self.init_delay_sec, self.max_delay_sec = init_delay_sec, max_delay_sec
        self.origin_robot_id = rid              # until the robot says otherwise
        self.first_heartbeat_received = False   # state "silent"
        self.init_time = self.latest_time_stamp = now
        self.last_keyframe_received = self.last_keyframe_sent = self.last_match_sent = -1
        self.heartbeat_subscriber = node.create_subscription(
            UInt32, HEARTBEAT_TOPIC.format(rid=rid), self.heartbeat_callback, _QUEUE_DEPTH)

    def heartbeat_callback(self, msg):
        """A beat carries the id of the sender's pose-graph origin.  The FIRST beat restarts the
        warm-up clock (neighbor_monitor.py:33-35)."""
        stamp = time.time()
        self.origin_robot_id = msg.data
        self.latest_time_stamp = stamp
        if not self.first_heartbeat_received:
            self.first_heartbeat_received = True
            self.init_time = time.time()

    def _warmed_up(self, now):
        return now - self.init_time > self.init_delay_sec

    def _fresh(self, now):
        return now - self.latest_time_stamp < self.max_delay_sec

    def is_alive(self):
        """Truthy iff a beat was seen, the warm-up is over and the last beat is recent.  With
        monitoring DISABLED the reference evaluates a bare `True` expression and falls off the
        end (neighbor_monitor.py:52-53): the result is None, i.e. falsy.  Callers only test
        truthiness, and the golden traces contain that behaviour, so it is kept."""
        if not self.is_enabled:
            return None
        now = time.time()
        return self.first_heartbeat_received and self._warmed_up(now) and self._fresh(now)

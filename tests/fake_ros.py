"""A minimal in-process stand-in for the ROS 2 node API used by the loop-closure surface:
publishers/subscriptions over a shared bus with ROS name resolution (relative names are
prefixed by the node namespace), manual timers, a parameter dict and a recording logger."""


class _Value(object):
    def __init__(self, v):
        self.value = v


class _Logger(object):
    def __init__(self):
        self.lines = []

    def info(self, m):
        self.lines.append(("info", m))

    def debug(self, m):
        self.lines.append(("debug", m))

    def error(self, m):
        self.lines.append(("error", m))

    warn = info


class FakeBus(object):
    def __init__(self):
        self.subs = {}
        self.log = []          # (topic, msg) of everything published
        self.queue = []
        self.deferred = False  # True: deliveries wait for spin()

    def subscribe(self, topic, cb):
        self.subs.setdefault(topic, []).append(cb)

    def publish(self, topic, msg):
        self.log.append((topic, msg))
        if self.deferred:
            self.queue.append((topic, msg))
        else:
            for cb in list(self.subs.get(topic, [])):
                cb(msg)

    def spin(self):
        while self.queue:
            topic, msg = self.queue.pop(0)
            for cb in list(self.subs.get(topic, [])):
                cb(msg)

    def published(self, topic):
        return [m for t, m in self.log if t == topic]


class _Publisher(object):
    def __init__(self, bus, topic):
        self.bus, self.topic = bus, topic

    def publish(self, msg):
        self.bus.publish(self.topic, msg)


class _Timer(object):
    def __init__(self, period, cb):
        self.period, self.cb = period, cb

    def fire(self):
        self.cb()


class FakeNode(object):
    def __init__(self, bus, namespace, parameters):
        self.bus = bus
        self.namespace = namespace.rstrip("/")
        self.parameters = dict(parameters)
        self.timers = []
        self.logger = _Logger()

    def resolve(self, topic):
        return topic if topic.startswith("/") else f"{self.namespace}/{topic}"

    def create_publisher(self, msg_type, topic, depth):
        return _Publisher(self.bus, self.resolve(topic))

    def create_subscription(self, msg_type, topic, cb, depth):
        self.bus.subscribe(self.resolve(topic), cb)
        return (topic, cb)

    def create_timer(self, period, cb, clock=None):
        t = _Timer(period, cb)
        self.timers.append(t)
        return t

    def get_parameter(self, name):
        return _Value(self.parameters[name])

    def get_logger(self):
        return self.logger

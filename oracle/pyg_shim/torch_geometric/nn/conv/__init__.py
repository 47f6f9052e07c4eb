from .message_passing import MessagePassing  # noqa: F401
from .edge_conv import EdgeConv  # noqa: F401
from .sage_conv import SAGEConv  # noqa: F401

from graph_tool import Graph, Vertex, Edge  # noqa: F401
